// rtk_capi.hip -- the C-ABI of librtk_amd.so: the nine rtk.h entry points plus the
// additive batch/device surface of rtk_amd.h. Thin: argument checks, residency cache,
// HIP plumbing. No intersection arithmetic lives here and there is no CPU fallback:
// every trace call runs the HIP kernels of rtk_trace.hip or fails with an error.
#include "rtk_dev.h"

#include <algorithm>
#include "rtk_layout_check.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <atomic>
#include <mutex>
#include <unordered_map>

// ---------------------------------------------------------------------------- errors

static thread_local char g_error[512] = "";

void rtk_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_error, sizeof(g_error), fmt, ap);
	va_end(ap);
}

extern "C" const char *rtk_amd_last_error(void) { return g_error; }


extern "C" int rtk_amd_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { rtk_set_error("hipGetDeviceCount failed: no HIP device or driver"); return 0; }
	return n;
}

extern "C" int rtk_amd_set_device(int device)
{
	RTK_HIP_CHECK(hipSetDevice(device), RTK_AMD_ERR_NO_DEVICE);
	return RTK_AMD_OK;
}

// ---------------------------------------------------------------------------- scenes

extern "C" rtk_dev_scene *rtk_dev_scene_upload(const rtk_scene *scene)
{
	HostBvh h;
	if (rtk_blob_to_host_bvh(scene, scene ? (size_t)scene->size_in_bytes : 0, &h) != RTK_AMD_OK) return nullptr;
	return rtk_dev_scene_from_host_bvh(h);
}

extern "C" rtk_dev_scene *rtk_dev_scene_upload_buffer(const void *blob, size_t blob_bytes)
{
	HostBvh h;
	if (!blob || blob_bytes < sizeof(rtk_scene)) { rtk_set_error("scene blob: buffer smaller than the header"); return nullptr; }
	if (rtk_blob_to_host_bvh(static_cast<const rtk_scene *>(blob), blob_bytes, &h) != RTK_AMD_OK) return nullptr;
	return rtk_dev_scene_from_host_bvh(h);
}

void rtk_export_forget(const rtk_dev_scene *ds);   // rtk_build.hip

extern "C" void rtk_dev_scene_free(rtk_dev_scene *ds)
{
	if (!ds) return;
	rtk_export_forget(ds);
	for (void *p : ds->allocs) (void)hipFree(p);
	for (LaunchScratch *s : ds->scratch) rtk_scratch_free(s);
	delete ds;
}

extern "C" int rtk_dev_scene_get_info(const rtk_dev_scene *ds, rtk_dev_scene_info *info)
{
	if (!ds || !info) { rtk_set_error("rtk_dev_scene_get_info: NULL argument"); return RTK_AMD_ERR_BAD_ARG; }
	info->num_triangles = ds->view.num_tris;
	info->num_meshes = ds->mesh_base.empty() ? 0 : ds->mesh_base.size() - 1;
	info->num_nodes = ds->view.num_nodes;
	info->node_bytes = (uint64_t)ds->view.num_nodes * sizeof(DevNode);
	info->triangle_bytes = (uint64_t)ds->view.num_tris * sizeof(DevTri);
	info->total_device_bytes = ds->total_bytes;
	info->max_depth = ds->max_depth;
	info->stack_entries = ds->stack_entries;
	info->build_ms = ds->build_ms;
	return RTK_AMD_OK;
}

extern "C" int rtk_dev_scene_mesh_base(const rtk_dev_scene *ds, uint64_t *out, size_t capacity)
{
	if (!ds || !out || capacity < ds->mesh_base.size()) { rtk_set_error("rtk_dev_scene_mesh_base: bad argument"); return RTK_AMD_ERR_BAD_ARG; }
	memcpy(out, ds->mesh_base.data(), ds->mesh_base.size() * sizeof(uint64_t));
	return (int)ds->mesh_base.size();
}

extern "C" long long rtk_dev_scene_primitive_order(const rtk_dev_scene *ds, uint32_t *out, size_t capacity)
{
	if (!ds || !out || capacity < ds->view.num_tris) { rtk_set_error("rtk_dev_scene_primitive_order: bad argument"); return RTK_AMD_ERR_BAD_ARG; }
	if (ds->view.num_tris == 0) return 0;
	// DevTri.prim is the 4th dword of each 48-byte record
	if (hipMemcpy2D(out, 4, reinterpret_cast<const char *>(ds->view.tris) + 12, sizeof(DevTri), 4, ds->view.num_tris, hipMemcpyDeviceToHost) != hipSuccess) {
		rtk_set_error("rtk_dev_scene_primitive_order: copy failed: %s", hipGetErrorString(hipGetLastError()));
		return RTK_AMD_ERR_HIP;
	}
	return (long long)ds->view.num_tris;
}

// ---------------------------------------------------------------------------- batches

extern "C" int rtk_dev_trace_rays(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_trace_opts *opts, void *stream)
{
	return rtk_launch_trace(ds, d_rays, n, d_hits, nullptr, opts, (hipStream_t)stream, false, nullptr);
}

extern "C" int rtk_dev_trace_rays_any(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	uint8_t *d_occluded, const rtk_trace_opts *opts, void *stream)
{
	return rtk_launch_trace(ds, d_rays, n, nullptr, d_occluded, opts, (hipStream_t)stream, true, nullptr);
}

extern "C" int rtk_dev_trace_rays_filtered(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_dev_filter *filter, const rtk_trace_opts *opts, void *stream)
{
	return rtk_launch_trace(ds, d_rays, n, d_hits, nullptr, opts, (hipStream_t)stream, false, nullptr, filter);
}

extern "C" int rtk_dev_trace_rays_any_filtered(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	uint8_t *d_occluded, const rtk_dev_filter *filter, const rtk_trace_opts *opts, void *stream)
{
	return rtk_launch_trace(ds, d_rays, n, nullptr, d_occluded, opts, (hipStream_t)stream, true, nullptr, filter);
}

extern "C" int rtk_dev_trace_rays_counted(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_trace_opts *opts, rtk_trace_counters *out)
{
	if (!out) { rtk_set_error("rtk_dev_trace_rays_counted: NULL counters"); return RTK_AMD_ERR_BAD_ARG; }
	return rtk_launch_trace(ds, d_rays, n, d_hits, nullptr, opts, nullptr, false, out);
}

extern "C" int rtk_dev_trace_rays_any_counted(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	uint8_t *d_occluded, const rtk_trace_opts *opts, rtk_trace_counters *out)
{
	if (!out) { rtk_set_error("rtk_dev_trace_rays_any_counted: NULL counters"); return RTK_AMD_ERR_BAD_ARG; }
	return rtk_launch_trace(ds, d_rays, n, nullptr, d_occluded, opts, nullptr, true, out);
}

extern "C" int rtk_dev_trace_rays_packet_counted(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n,
	rtk_hit_record *d_hits, const rtk_trace_opts *opts, rtk_packet_counters *out)
{
	if (!out) { rtk_set_error("rtk_dev_trace_rays_packet_counted: NULL counters"); return RTK_AMD_ERR_BAD_ARG; }
	return rtk_launch_trace(ds, d_rays, n, d_hits, nullptr, opts, nullptr, false, nullptr, nullptr, nullptr, nullptr, 0, out);
}

extern "C" int rtk_dev_detect_image(const rtk_dev_scene *ds, const rtk_ray *d_rays, size_t n, uint32_t *width, uint32_t *height, void *stream)
{
	if (!ds || !width || !height || (!d_rays && n)) { rtk_set_error("rtk_dev_detect_image: NULL argument"); return RTK_AMD_ERR_BAD_ARG; }
	return rtk_detect_image(ds, d_rays, n, (hipStream_t)stream, width, height);
}

extern "C" int rtk_dev_trace_status(const rtk_dev_scene *ds, void *stream)
{
	return rtk_trace_status(ds, (hipStream_t)stream);
}

extern "C" int rtk_dev_expand_hits(const rtk_dev_scene *ds, const rtk_hit_record *d_records, size_t n,
	rtk_hit *d_hits, uint8_t *d_mask, void *stream)
{
	return rtk_launch_expand(ds, d_records, n, d_hits, d_mask, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- residency cache
//
// Host-pointer calls (rtk_trace_ray[s], the reference's own signatures) find the device copy of a blob through its
// address and the calling thread's current device. The address alone is not an identity: rtk_finish_build_to writes
// into caller memory that the caller releases with free(), so a different blob can later live at the same address.
//   * Every way this library itself puts a blob at an address (rtk_finish_build[_to] with either builder, rtk_free_scene)
//     drops or replaces the entry.
//   * When a blob is uploaded, EVERY 4 KB stripe of it is hashed (one pass, ~1 ns per 8 bytes). A lookup re-checks the
//     64-byte header and the root node (a different scene of another size or shape is caught at once) plus ONE stripe, a
//     different one each time, in rotation: a caller that rewrites a blob in place behind the library's back -- one moved
//     vertex -- is found out within (size / 4 KB) lookups instead of never, at ~0.2 us per call instead of a re-hash of
//     up to 100 MB. rtk_amd_forget_scene is the immediate, documented way (include/rtk_amd.h).

namespace {

const size_t STRIPE = 4096;

uint64_t hash_words(const void *data, size_t n, uint64_t h)
{
	const unsigned char *p = static_cast<const unsigned char *>(data);
	size_t i = 0;
	for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x9e3779b97f4a7c15ull; h ^= h >> 29; }
	for (; i < n; i++) { h = (h ^ p[i]) * 0x100000001b3ull; }
	return h;
}

uint64_t head_hash(const rtk_scene *scene)
{
	const uint64_t size = scene->size_in_bytes;
	uint64_t h = hash_words(scene, sizeof(rtk_scene), 0xcbf29ce484222325ull);
	if (size >= 256 && size <= ((uint64_t)1 << 48)) h = hash_words(reinterpret_cast<const char *>(scene) + 128, 128, h);   // root node
	return h;
}

struct CacheEntry {
	rtk_dev_scene *ds = nullptr;
	uint64_t size = 0, head = 0;
	std::vector<uint64_t> stripes;     // hash of every STRIPE bytes of the blob as it was uploaded
	size_t next = 0;                   // the stripe the next lookup re-checks
};
struct CacheKey {
	const rtk_scene *scene; int device;
	bool operator==(const CacheKey &o) const { return scene == o.scene && device == o.device; }
};
struct CacheKeyHash { size_t operator()(const CacheKey &k) const { return std::hash<const void *>()(k.scene) ^ ((size_t)k.device * 0x9e3779b97f4a7c15ull); } };
std::mutex g_cache_mutex;
std::unordered_map<CacheKey, CacheEntry, CacheKeyHash> g_cache;
thread_local bool t_fatal = false;     // the last resident() of this thread failed for good (invalid scene / no device)

void fill_entry(CacheEntry &e, const rtk_scene *scene, rtk_dev_scene *ds)
{
	e.ds = ds;
	e.size = scene->size_in_bytes;
	e.head = head_hash(scene);
	e.next = 0;
	e.stripes.clear();
	if (e.size >= 256 && e.size <= ((uint64_t)1 << 48)) {
		const char *b = reinterpret_cast<const char *>(scene);
		for (uint64_t at = 0; at < e.size; at += STRIPE) e.stripes.push_back(hash_words(b + at, (size_t)std::min<uint64_t>(STRIPE, e.size - at), at));
	}
}

int current_device()
{
	int d = 0;
	if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
	return d;
}

rtk_dev_scene *resident(const rtk_scene *scene)
{
	const CacheKey key = { scene, current_device() };
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	auto it = g_cache.find(key);
	if (it != g_cache.end()) {
		CacheEntry &e = it->second;
		bool same = e.size == scene->size_in_bytes && e.head == head_hash(scene);
		if (same && !e.stripes.empty()) {
			const size_t k = e.next;
			e.next = (k + 1) % e.stripes.size();
			const uint64_t at = (uint64_t)k * STRIPE;
			same = hash_words(reinterpret_cast<const char *>(scene) + at, (size_t)std::min<uint64_t>(STRIPE, e.size - at), at) == e.stripes[k];
		}
		if (same) return e.ds;
		rtk_dev_scene_free(e.ds);                                     // another blob lives at this address now
		g_cache.erase(it);
	}
	// what kind of failure a NULL is: a blob that does not validate or a machine without a usable GPU will fail the same
	// way on every call ("fatal" for rtk_trace_ray, which has no error channel); running out of memory may not
	t_fatal = false;
	HostBvh h;
	if (rtk_blob_to_host_bvh(scene, (size_t)scene->size_in_bytes, &h) != RTK_AMD_OK) { t_fatal = true; return nullptr; }
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); t_fatal = true; rtk_set_error("no usable HIP device"); return nullptr; }
	rtk_dev_scene *ds = rtk_dev_scene_from_host_bvh(h);
	if (ds) fill_entry(g_cache[key], scene, ds);
	return ds;
}

} // namespace

bool rtk_last_failure_is_fatal() { return t_fatal; }

void rtk_cache_adopt(const rtk_scene *scene, rtk_dev_scene *ds)
{
	const CacheKey key = { scene, ds ? ds->device : current_device() };
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	// nothing that lived at this address before is valid any more, on any device
	for (auto it = g_cache.begin(); it != g_cache.end();) {
		if (it->first.scene == scene) { rtk_dev_scene_free(it->second.ds); it = g_cache.erase(it); } else ++it;
	}
	fill_entry(g_cache[key], scene, ds);
}

extern "C" void rtk_amd_forget_scene(const rtk_scene *scene)
{
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	for (auto it = g_cache.begin(); it != g_cache.end();) {
		if (it->first.scene == scene) { rtk_dev_scene_free(it->second.ds); it = g_cache.erase(it); } else ++it;
	}
}

// ---------------------------------------------------------------------------- host-pointer tracing
//
// Each host thread owns a stream, pinned staging memory and device buffers for one chunk of rays, all
// created on first use and kept: a call allocates nothing in steady state (the first version paid four
// hipMalloc + four hipFree per rtk_trace_ray). Batches larger than a chunk stream through in pieces.

namespace {

const size_t ZERO_COPY_RAYS = 2048;            // pieces up to this size are read and written in place by the kernels

// this thread's stream is going away: every cached scene drops the scratch set it made for it
void drop_stream_everywhere(hipStream_t stream)
{
	if (!stream) return;
	(void)hipStreamSynchronize(stream);
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	for (auto &kv : g_cache) rtk_scene_drop_stream(kv.second.ds, stream);
}

struct HostCtx {
	int device = -1;
	hipStream_t stream = nullptr;
	size_t cap = 0;                            // rays
	char *pinned = nullptr;                    // [rays | records | hits | mask | after]
	char *dev = nullptr;
	rtk_ray *h_rays = nullptr, *d_rays = nullptr;
	rtk_hit_record *h_rec = nullptr, *d_rec = nullptr;
	rtk_hit *h_hits = nullptr, *d_hits = nullptr;
	uint8_t *h_mask = nullptr, *d_mask = nullptr;
	rtk_hit_record *h_after = nullptr, *d_after = nullptr;
	unsigned long long *h_status = nullptr;    // pinned: the expand kernel of a zero-copy piece mirrors the launch-error word here
	bool status_mirrored = false;              // ... for the piece in flight
	uint32_t ticket = 0, ticket_in_flight = 0; // pieces of one workgroup (<= 256 rays) are waited for by polling h_status for their ticket
	// candidate lists for host-callback filters: CAND_SLOTS records in all, k = CAND_SLOTS / rays of them per ray
	rtk_hit_record *h_cand = nullptr, *d_cand = nullptr;
	rtk_hit *h_cand_hits = nullptr, *d_cand_hits = nullptr;
	uint32_t *h_cand_count = nullptr, *d_cand_count = nullptr;
	bool cand_ready = false;

	void release()
	{
		if (pinned) (void)hipHostFree(pinned);
		if (dev) (void)hipFree(dev);
		pinned = dev = nullptr;
		cap = 0;
		if (h_cand) (void)hipHostFree(h_cand);
		if (h_cand_hits) (void)hipHostFree(h_cand_hits);
		if (h_cand_count) (void)hipHostFree(h_cand_count);
		if (d_cand) (void)hipFree(d_cand);
		if (d_cand_hits) (void)hipFree(d_cand_hits);
		if (d_cand_count) (void)hipFree(d_cand_count);
		h_cand = d_cand = nullptr; h_cand_hits = d_cand_hits = nullptr; h_cand_count = d_cand_count = nullptr;
		cand_ready = false;
	}
	bool ensure_candidates();
	~HostCtx()
	{
		release();
		if (stream) { drop_stream_everywhere(stream); (void)hipStreamDestroy(stream); }
	}
	bool ensure(size_t n)
	{
		int cur = 0;
		if (hipGetDevice(&cur) != hipSuccess) { rtk_set_error("no usable HIP device: %s", hipGetErrorString(hipGetLastError())); return false; }
		if (cur != device) {                   // the thread moved to another GPU
			release();
			if (stream) { drop_stream_everywhere(stream); (void)hipStreamDestroy(stream); }
			stream = nullptr;
			device = cur;
		}
		if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
			rtk_set_error("hipStreamCreate failed: %s", hipGetErrorString(hipGetLastError()));
			stream = nullptr;
			return false;
		}
		if (n <= cap) return true;
		release();
		size_t want = 64;
		while (want < n) want <<= 1;
		const size_t per_ray = sizeof(rtk_ray) + 2 * sizeof(rtk_hit_record) + sizeof(rtk_hit) + 4 /* mask, kept 4-aligned */;
		if (hipHostMalloc((void **)&pinned, want * per_ray + 64, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&dev, want * per_ray + 64) != hipSuccess) {
			rtk_set_error("rtk_trace_rays: staging allocation failed (%zu rays): %s", want, hipGetErrorString(hipGetLastError()));
			release();
			return false;
		}
		cap = want;
		auto carve = [&](char *base) {
			size_t off = 0;
			auto take = [&](size_t bytes) { char *p = base + off; off += bytes; return p; };
			rtk_ray *r = (rtk_ray *)take(want * sizeof(rtk_ray));
			rtk_hit_record *rec = (rtk_hit_record *)take(want * sizeof(rtk_hit_record));
			rtk_hit_record *aft = (rtk_hit_record *)take(want * sizeof(rtk_hit_record));
			rtk_hit *hh = (rtk_hit *)take(want * sizeof(rtk_hit));
			uint8_t *m = (uint8_t *)take(want * 4);
			unsigned long long *st = (unsigned long long *)take(64);
			if (base == pinned) { h_rays = r; h_rec = rec; h_after = aft; h_hits = hh; h_mask = m; h_status = st; *st = 0ull; }
			else { d_rays = r; d_rec = rec; d_after = aft; d_hits = hh; d_mask = m; }
		};
		carve(pinned);
		carve(dev);
		return true;
	}
};

const size_t CAND_SLOTS = (size_t)1 << 16;       // candidate records per round (4.4 MB of full hits)
const size_t CAND_RAYS = (size_t)1 << 14;        // rays per round of a host-callback filter call: at least 4 candidates each

bool HostCtx::ensure_candidates()
{
	if (cand_ready) return true;
	const bool ok = hipHostMalloc((void **)&h_cand, CAND_SLOTS * sizeof(rtk_hit_record), hipHostMallocDefault) == hipSuccess &&
		hipHostMalloc((void **)&h_cand_hits, CAND_SLOTS * sizeof(rtk_hit), hipHostMallocDefault) == hipSuccess &&
		hipHostMalloc((void **)&h_cand_count, CAND_RAYS * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess &&
		hipMalloc((void **)&d_cand, CAND_SLOTS * sizeof(rtk_hit_record)) == hipSuccess &&
		hipMalloc((void **)&d_cand_hits, CAND_SLOTS * sizeof(rtk_hit)) == hipSuccess &&
		hipMalloc((void **)&d_cand_count, CAND_RAYS * sizeof(uint32_t)) == hipSuccess;
	if (!ok) { rtk_set_error("rtk_trace_rays_filter: candidate buffers: %s", hipGetErrorString(hipGetLastError())); return false; }
	cand_ready = true;
	return true;
}

thread_local HostCtx t_ctx;

// One piece (n <= chunk capacity), first half: rays up, trace (optionally restricted to candidates after h_after),
// expand, hits and mask on their way down -- everything enqueued on the context's stream, nothing waited for.
bool enqueue_piece(rtk_dev_scene *ds, HostCtx &c, const rtk_ray *rays, size_t n, bool want_hits, bool with_after, const rtk_trace_opts *opts = nullptr)
{
	memcpy(c.h_rays, rays, n * sizeof(rtk_ray));
	// Small pieces skip the copy engines altogether: the pinned staging memory is visible to the device, so the kernels
	// read the rays from it and write hits and mask into it over PCIe themselves -- two launches and one synchronisation
	// instead of those plus four copies (rtk_trace_ray: 74 us -> 50 us; 34 us with the status word and the ticket below, profiles/r02_single_ray_latency.log).
	const bool zero_copy = n <= ZERO_COPY_RAYS && !with_after;
	const rtk_ray *d_rays = zero_copy ? c.h_rays : c.d_rays;
	if (!zero_copy && hipMemcpyAsync(c.d_rays, c.h_rays, n * sizeof(rtk_ray), hipMemcpyHostToDevice, c.stream) != hipSuccess) { rtk_set_error("rtk_trace_rays: H2D copy failed"); return false; }
	rtk_dev_filter f;
	memset(&f, 0, sizeof(f));
	f.struct_size = sizeof(f);
	if (with_after) {
		if (hipMemcpyAsync(c.d_after, c.h_after, n * sizeof(rtk_hit_record), hipMemcpyHostToDevice, c.stream) != hipSuccess) { rtk_set_error("rtk_trace_rays: H2D copy failed"); return false; }
		f.d_after = c.d_after;
	}
	if (rtk_launch_trace(ds, d_rays, n, c.d_rec, nullptr, opts, c.stream, false, nullptr, with_after ? &f : nullptr) != RTK_AMD_OK) return false;
	c.status_mirrored = false;
	if (zero_copy) {
		c.ticket_in_flight = 0;
		if (n <= 256) { c.ticket = c.ticket == 0xffffffffu ? 1u : c.ticket + 1u; c.ticket_in_flight = c.ticket; }
		if (rtk_launch_expand(ds, c.d_rec, n, want_hits ? c.h_hits : nullptr, c.h_mask, c.stream, c.h_status, c.ticket_in_flight) != RTK_AMD_OK) return false;
		c.status_mirrored = true;
	} else {
		if (rtk_launch_expand(ds, c.d_rec, n, want_hits ? c.d_hits : nullptr, c.d_mask, c.stream) != RTK_AMD_OK) return false;
		bool ok = hipMemcpyAsync(c.h_mask, c.d_mask, n, hipMemcpyDeviceToHost, c.stream) == hipSuccess;
		if (want_hits) ok = ok && hipMemcpyAsync(c.h_hits, c.d_hits, n * sizeof(rtk_hit), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
		if (!ok) { rtk_set_error("rtk_trace_rays: D2H copy failed: %s", hipGetErrorString(hipGetLastError())); return false; }
	}
	return true;
}

// Second half: wait for the piece, then hand its results to the caller's arrays. Returns the hits, (size_t)-1 on error.
size_t finish_piece(rtk_dev_scene *ds, HostCtx &c, size_t n, rtk_hit *hits, uint8_t *hit_mask)
{
	if (c.status_mirrored) {
		// the error word came down with the results: one synchronisation, no transfer (the slow path below reports and resets it)
		bool arrived = false;
		if (c.ticket_in_flight) {
			// a lone workgroup signs off with the ticket after its results (release store at system scope): poll for it -- a wait
			// on the stream costs the runtime's wake-up latency on top -- and fall back to the stream after ~2 ms without it
			const volatile unsigned long long *w = c.h_status;
			const auto t0 = std::chrono::steady_clock::now();
			for (uint32_t spin = 0;; spin++) {
				if ((uint32_t)(__atomic_load_n(w, __ATOMIC_ACQUIRE) >> 32) == c.ticket_in_flight) { arrived = true; break; }
				if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
			}
		}
		if (!arrived && hipStreamSynchronize(c.stream) != hipSuccess) { rtk_set_error("rtk_trace_rays: %s", hipGetErrorString(hipGetLastError())); return (size_t)-1; }
		if ((*c.h_status & 0xffffffffull) != 0ull && rtk_trace_status(ds, c.stream) != RTK_AMD_OK) return (size_t)-1;
	} else if (rtk_trace_status(ds, c.stream) != RTK_AMD_OK) return (size_t)-1;      // synchronises the stream
	size_t count = 0;
	for (size_t i = 0; i < n; i++) {
		if (c.h_mask[i]) { count++; if (hits) hits[i] = c.h_hits[i]; }           // misses stay untouched (rtk.c:571-576)
		if (hit_mask) hit_mask[i] = c.h_mask[i];
	}
	return count;
}

// rtk_trace_ray's own path: one ray, ONE launch (rtk_trace_one_kernel walks the ray's frontier breadth first with a whole wave
// and writes the full rtk_hit, the mask and the ticket into the pinned staging memory itself). Returns the hits (0 / 1),
// (size_t)-1 on error, (size_t)-2 when the kernel reported "not done" (a frontier that does not fit LDS): the caller then takes
// the batch path.
size_t trace_one(rtk_dev_scene *ds, HostCtx &c, const rtk_ray *ray, rtk_hit *hit, uint8_t *hit_mask)
{
	c.h_rays[0] = *ray;
	c.ticket = c.ticket == 0xffffffffu ? 1u : c.ticket + 1u;
	const int launched = rtk_launch_trace_one(ds, c.h_rays, c.h_hits, c.h_mask, c.stream, c.h_status, c.ticket);
	if (launched == RTK_AMD_ERR_UNSUPPORTED) return (size_t)-2;          // (a scene beyond the one-ray kernel's 32-bit offsets: the batch path diagnoses it)
	if (launched != RTK_AMD_OK) return (size_t)-1;
	const volatile unsigned long long *w = c.h_status;
	const auto t0 = std::chrono::steady_clock::now();
	bool arrived = false;
	for (uint32_t spin = 0;; spin++) {
		if ((uint32_t)(__atomic_load_n(w, __ATOMIC_ACQUIRE) >> 32) == c.ticket) { arrived = true; break; }
		if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
	}
	if (!arrived) {
		if (hipStreamSynchronize(c.stream) != hipSuccess) { rtk_set_error("rtk_trace_ray: %s", hipGetErrorString(hipGetLastError())); return (size_t)-1; }
		if ((uint32_t)(*c.h_status >> 32) != c.ticket) { rtk_set_error("rtk_trace_ray: the kernel finished without its ticket"); return (size_t)-1; }
	}
	if ((*c.h_status & 0xffffffffull) == 2ull) return (size_t)-2;
	const uint8_t m = c.h_mask[0];
	if (m && hit) *hit = c.h_hits[0];
	if (hit_mask) *hit_mask = m;
	return m ? 1 : 0;
}

// The look of rtk_dev_trace_rays for an un-announced image (rtk_trace.hip, k_detect_row / k_detect_check), on host rays: the step
// from ray to ray (origin and direction) is regular and jumps at the same distance every time. A batch of host rays is cut into
// pieces for the staging buffers; an image is cut into bands of whole 64-pixel rows so that every piece is an image the packet
// kernels take.
bool host_step_jumps(const rtk_ray *rays, size_t i)
{
	const float *a = reinterpret_cast<const float *>(rays + i - 1), *b = reinterpret_cast<const float *>(rays + i), *c = reinterpret_cast<const float *>(rays + i + 1);
	float m = 0.0f, dmax = 0.0f;
	for (int k = 0; k < 6; k++) {
		const float s0 = b[k] - a[k], s1 = c[k] - b[k];
		m = fabsf(s0) > m ? fabsf(s0) : m;
		dmax = fabsf(s1 - s0) > dmax ? fabsf(s1 - s0) : dmax;
	}
	return !(dmax <= 8.0f * m);
}

void host_detect_image(const rtk_ray *rays, size_t n, uint32_t *w_out, uint32_t *h_out)
{
	*w_out = *h_out = 0u;
	if (n < 4 || n > 0x40000000ull) return;
	const size_t limit = n < ((size_t)1 << 17) ? n : ((size_t)1 << 17);
	size_t first = 0;
	for (size_t i = 1; i + 1 < limit; i++) if (host_step_jumps(rays, i)) { first = i; break; }
	const size_t w = first + 1;
	if (!first || w < 64 || (n % w) != 0 || n / w < 2 || n / w > 0xffffffffull) return;
	const size_t rows = n / w, stride = rows > 256 ? rows / 256 : 1;
	for (size_t k = 0; k < 256; k++) {
		const size_t r = k * stride;
		if (r + 1 >= rows) break;
		if (!host_step_jumps(rays, (r + 1) * w - 1)) return;
		for (size_t q = 1; q < 4; q++) if (host_step_jumps(rays, r * w + q * (w / 4))) return;
	}
	*w_out = (uint32_t)w;
	*h_out = (uint32_t)rows;
}

std::atomic<int> g_test_fail_calls{0};
// Fault injection for the tests of the per-ray calls' failure reporting. Not in the installed header; a no-op unless the process
// was started with RTK_AMD_TEST_HOOKS=1 (read once): no code of a production host can make its traces fail through it.
extern "C" void rtk_amd_test_fail_next_calls(int calls)
{
	static const bool armed = getenv("RTK_AMD_TEST_HOOKS") && !strcmp(getenv("RTK_AMD_TEST_HOOKS"), "1");
	if (armed) g_test_fail_calls.store(calls > 0 ? calls : 0);
}
thread_local HostCtx t_ctx2;                    // second staging set (own stream) for pipelined host-pointer batches
const size_t PIPE_CHUNK_RAYS = (size_t)1 << 15; // rays per piece when a batch is pipelined

} // namespace

extern "C" size_t rtk_trace_rays(const rtk_scene *scene, const rtk_ray *rays, size_t n, rtk_hit *hits, uint8_t *hit_mask)
{
	if (!scene || (!rays && n)) { rtk_set_error("rtk_trace_rays: NULL argument"); return (size_t)-1; }
	if (n == 0) return 0;
	t_fatal = false;
	if (g_test_fail_calls.load(std::memory_order_relaxed) > 0 && g_test_fail_calls.fetch_sub(1) > 0) {
		rtk_set_error("rtk_trace_rays: injected failure (rtk_amd_test_fail_next_calls)");      // tests of the per-ray calls' failure reporting
		return (size_t)-1;
	}
	rtk_dev_scene *ds = resident(scene);
	if (!ds) return (size_t)-1;
	if (n < 2 * PIPE_CHUNK_RAYS) {
		HostCtx &c = t_ctx;
		if (!c.ensure(n)) return (size_t)-1;
		static const int one_default = getenv("RTK_AMD_ONE_RAY_KERNEL") ? atoi(getenv("RTK_AMD_ONE_RAY_KERNEL")) : 1;
		if (n == 1 && one_default != 0) {
			const size_t r = trace_one(ds, c, rays, hits, hit_mask);
			if (r != (size_t)-2) return r;
		}
		if (!enqueue_piece(ds, c, rays, n, hits != nullptr, false)) return (size_t)-1;
		return finish_piece(ds, c, n, hits, hit_mask);
	}
	// Two staging sets on two streams: while the GPU works on one piece the host copies the previous piece's results out
	// and the next piece's rays in (both single-threaded memcpy-speed work that used to sit between the launches).
	// An image (recognised by its regular step, as rtk_dev_trace_rays does for device rays) goes in bands of whole 64-pixel rows, each
	// announced to the launch as the image it is: the packet kernels instead of one ray per lane.
	size_t PIPE_CHUNK = PIPE_CHUNK_RAYS;
	rtk_trace_opts band_opts;
	const rtk_trace_opts *piece_opts = nullptr;
	{
		uint32_t iw = 0, ih = 0;
		static const bool detect = !(getenv("RTK_AMD_DETECT_IMAGE") && atoi(getenv("RTK_AMD_DETECT_IMAGE")) == 0);
		if (detect) host_detect_image(rays, n, &iw, &ih);
		if (iw >= 128u && (iw % 64u) == 0u && (ih % 64u) == 0u && (size_t)iw * 64u <= ((size_t)1 << 21)) {
			size_t band_rows = 64;
			while ((band_rows * 2) * (size_t)iw <= ((size_t)1 << 18) && band_rows * 2 <= ih) band_rows *= 2;
			PIPE_CHUNK = band_rows * (size_t)iw;
			memset(&band_opts, 0, sizeof(band_opts));
			band_opts.struct_size = sizeof(band_opts);
			band_opts.image_width = iw;
			piece_opts = &band_opts;
		}
	}
	HostCtx *ctx[2] = { &t_ctx, &t_ctx2 };
	if (!ctx[0]->ensure(PIPE_CHUNK) || !ctx[1]->ensure(PIPE_CHUNK)) return (size_t)-1;
	size_t count = 0, piece = 0;
	size_t at_of[2] = { 0, 0 }, n_of[2] = { 0, 0 };
	bool busy[2] = { false, false }, failed = false;
	for (size_t at = 0; at < n && !failed; at += PIPE_CHUNK, piece++) {
		const int k = (int)(piece & 1u);
		if (busy[k]) {
			const size_t r = finish_piece(ds, *ctx[k], n_of[k], hits ? hits + at_of[k] : nullptr, hit_mask ? hit_mask + at_of[k] : nullptr);
			busy[k] = false;
			if (r == (size_t)-1) { failed = true; break; }
			count += r;
		}
		const size_t m = n - at < PIPE_CHUNK ? n - at : PIPE_CHUNK;
		if (piece_opts) band_opts.image_height = (uint32_t)(m / band_opts.image_width);      // (the last band may be lower; still whole 64-pixel rows)
		if (!enqueue_piece(ds, *ctx[k], rays + at, m, hits != nullptr, false, piece_opts)) { failed = true; break; }
		busy[k] = true; at_of[k] = at; n_of[k] = m;
	}
	// drain in submission order (the older piece first)
	for (int j = 0; j < 2; j++) {
		const int k = (int)((piece + (size_t)j) & 1u);
		if (!busy[k]) continue;
		const size_t r = finish_piece(ds, *ctx[k], n_of[k], hits ? hits + at_of[k] : nullptr, hit_mask ? hit_mask + at_of[k] : nullptr);
		busy[k] = false;
		if (r == (size_t)-1) failed = true; else count += r;
	}
	return failed ? (size_t)-1 : count;
}

// Batch form of rtk_trace_ray_filter (rtk.h:117, 130; a stub in the reference, rtk.c:579-582). Semantics:
// for every ray the closest candidate hit the filter accepts. Candidates of a ray are offered in increasing
// (t, primitive id) order -- every candidate, including several at one and the same t -- until one is
// accepted or none is left. The callback runs on the host, so the batch goes in rounds: a launch collects
// the k closest candidates of every undecided ray (k = 65536 / rays in the round: 4 for a full round, up to 64
// for a single ray), the callback is asked about them in order, and rays whose k candidates were all rejected
// go into the next round restricted to what comes after the last one. Launches = rounds, not candidates: a
// single ray needs one launch per 64 rejected candidates.
extern "C" size_t rtk_trace_rays_filter(const rtk_scene *scene, const rtk_ray *rays, size_t n, rtk_hit *hits, uint8_t *hit_mask,
	rtk_filter_fn *filter, void *filter_user)
{
	if (!filter) return rtk_trace_rays(scene, rays, n, hits, hit_mask);
	if (!scene || (!rays && n)) { rtk_set_error("rtk_trace_rays_filter: NULL argument"); return (size_t)-1; }
	if (n == 0) return 0;
	t_fatal = false;
	rtk_dev_scene *ds = resident(scene);
	if (!ds) return (size_t)-1;
	HostCtx &c = t_ctx;
	if (!c.ensure(n < CAND_RAYS ? n : CAND_RAYS) || !c.ensure_candidates()) return (size_t)-1;
	size_t count = 0;
	std::vector<size_t> todo, next;
	std::vector<rtk_hit_record> after, next_after;
	rtk_dev_filter f;
	memset(&f, 0, sizeof(f));
	f.struct_size = sizeof(f);
	for (size_t at = 0; at < n; at += CAND_RAYS) {
		const size_t m = n - at < CAND_RAYS ? n - at : CAND_RAYS;
		todo.resize(m);
		for (size_t i = 0; i < m; i++) todo[i] = at + i;
		after.assign(m, rtk_hit_record{ 0.0f, 0.0f, 0.0f, RTK_PRIM_NONE });
		if (hit_mask) memset(hit_mask + at, 0, m);
		for (unsigned round = 0; !todo.empty(); round++) {
			if (round > (1u << 16)) { rtk_set_error("rtk_trace_rays_filter: a ray had more than 2^18 rejected candidates"); return (size_t)-1; }
			const size_t r = todo.size();
			size_t k = CAND_SLOTS / r;
			if (k > 64) k = 64;
			for (size_t q = 0; q < r; q++) { c.h_rays[q] = rays[todo[q]]; c.h_after[q] = after[q]; }
			bool ok = hipMemcpyAsync(c.d_rays, c.h_rays, r * sizeof(rtk_ray), hipMemcpyHostToDevice, c.stream) == hipSuccess &&
				hipMemcpyAsync(c.d_after, c.h_after, r * sizeof(rtk_hit_record), hipMemcpyHostToDevice, c.stream) == hipSuccess;
			if (!ok) { rtk_set_error("rtk_trace_rays_filter: H2D copy failed"); return (size_t)-1; }
			f.d_after = c.d_after;
			// unused list slots stay "no primitive" so that the expand kernel leaves them alone
			if (hipMemsetAsync(c.d_cand, 0xff, r * k * sizeof(rtk_hit_record), c.stream) != hipSuccess) { rtk_set_error("rtk_trace_rays_filter: memset failed"); return (size_t)-1; }
			if (rtk_launch_trace(ds, c.d_rays, r, nullptr, nullptr, nullptr, c.stream, false, nullptr, &f, c.d_cand, c.d_cand_count, (uint32_t)k) != RTK_AMD_OK) return (size_t)-1;
			if (rtk_launch_expand(ds, c.d_cand, r * k, c.d_cand_hits, nullptr, c.stream) != RTK_AMD_OK) return (size_t)-1;
			ok = hipMemcpyAsync(c.h_cand, c.d_cand, r * k * sizeof(rtk_hit_record), hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
				hipMemcpyAsync(c.h_cand_hits, c.d_cand_hits, r * k * sizeof(rtk_hit), hipMemcpyDeviceToHost, c.stream) == hipSuccess &&
				hipMemcpyAsync(c.h_cand_count, c.d_cand_count, r * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream) == hipSuccess;
			if (!ok) { rtk_set_error("rtk_trace_rays_filter: D2H copy failed: %s", hipGetErrorString(hipGetLastError())); return (size_t)-1; }
			if (rtk_trace_status(ds, c.stream) != RTK_AMD_OK) return (size_t)-1;      // synchronises the stream
			next.clear();
			next_after.clear();
			for (size_t q = 0; q < r; q++) {
				const size_t i = todo[q];
				const uint32_t have = c.h_cand_count[q];
				bool accepted = false;
				for (uint32_t j = 0; j < have && !accepted; j++) {
					if (filter(filter_user, &rays[i], &c.h_cand_hits[q * k + j])) {
						accepted = true;
						count++;
						if (hits) hits[i] = c.h_cand_hits[q * k + j];
						if (hit_mask) hit_mask[i] = 1;
					}
				}
				if (!accepted && have == k) {                                      // the list was full: there may be more behind it
					next.push_back(i);
					next_after.push_back(c.h_cand[q * k + k - 1]);
				}
			}
			todo.swap(next);
			after.swap(next_after);
		}
	}
	return count;
}

// ---------------------------------------------------------------------------- rtk.h: trace

// What a per-ray call does when its batch of one failed. The signatures have no error channel (the reference's entry point
// cannot fail) and `false` means "miss" to a host that only knows rtk.h, so a failure is never silent:
//   * every failure prints one line on stderr (the first 8 of a process, then every 1024th) and sets rtk_amd_last_error();
//   * one that every later call would repeat -- no usable GPU, a scene that does not validate -- or a SECOND failure in a row
//     on the calling thread (a stream error that sticks looks like a transient one the first time) would turn every ray
//     into a wrong "miss": report and stop, unless the host opted into soft failures (RTK_AMD_SOFT_ERRORS: false +
//     rtk_amd_last_error(), still reported on stderr);
//   * a lone failure that may pass (out of memory, an interrupted launch) returns false.
static thread_local unsigned t_per_ray_failures = 0;
static std::atomic<unsigned long long> g_per_ray_failures_reported{0};

static bool per_ray_failure(const char *who)
{
	const unsigned in_a_row = ++t_per_ray_failures;
	const unsigned long long seen = g_per_ray_failures_reported.fetch_add(1);
	const bool fatal = rtk_last_failure_is_fatal() || in_a_row >= 2u;
	const bool soft = getenv("RTK_AMD_SOFT_ERRORS") != nullptr;
	if (seen < 8ull || (seen & 1023ull) == 0ull || (fatal && !soft))
		fprintf(stderr, "%s: FAILED, not a miss (%s)%s: %s\n", who, fatal ? (in_a_row >= 2u ? "second failure in a row" : "permanent") : "transient",
			seen >= 8ull ? " [further reports are rate-limited]" : "", g_error);
	if (fatal && !soft) abort();
	return false;
}

// The per-ray symbols on the calling thread (rtk_host_trace.cpp): see there why. RTK_AMD_PER_RAY=gpu sends them through the
// GPU again (a batch of one: rtk_trace_one_kernel, one launch + a ticket, ~25 us).
int rtk_host_trace_ray(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, const rtk_hit *after);
int rtk_host_trace_ray_filter(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, rtk_filter_fn *filter, void *user);

static std::atomic<int> g_per_ray_where{-1};           // -1: not decided yet (RTK_AMD_PER_RAY=gpu is read once)

extern "C" int rtk_amd_set_per_ray(int where)
{
	if (where != RTK_AMD_PER_RAY_HOST && where != RTK_AMD_PER_RAY_GPU) { rtk_set_error("rtk_amd_set_per_ray: unknown value %d", where); return RTK_AMD_ERR_BAD_ARG; }
	g_per_ray_where.store(where);
	return RTK_AMD_OK;
}

static bool per_ray_on_gpu()
{
	int w = g_per_ray_where.load(std::memory_order_relaxed);
	if (w < 0) {
		const char *e = getenv("RTK_AMD_PER_RAY");
		w = (e && !strcmp(e, "gpu")) ? RTK_AMD_PER_RAY_GPU : RTK_AMD_PER_RAY_HOST;
		g_per_ray_where.store(w);
	}
	return w == RTK_AMD_PER_RAY_GPU;
}

static bool injected_failure(const char *who)
{
	if (g_test_fail_calls.load(std::memory_order_relaxed) > 0 && g_test_fail_calls.fetch_sub(1) > 0) {
		rtk_set_error("%s: injected failure (rtk_amd_test_fail_next_calls)", who);
		t_fatal = false;
		return true;
	}
	return false;
}

// reference rtk.h:129 / rtk.c:543-577. Re-entrant, any thread, `*hit` untouched on a miss. Served on the calling thread from
// the caller's blob (< 1.5 us; profiles/r05_c_host_latency.log); throughput callers use rtk_trace_rays / rtk_dev_trace_rays,
// which run on the GPU and nowhere else.
extern "C" bool rtk_trace_ray(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit)
{
	if (!scene || !ray || !hit) { rtk_set_error("rtk_trace_ray: NULL argument"); return false; }
	if (!per_ray_on_gpu()) {
		if (injected_failure("rtk_trace_ray")) return per_ray_failure("rtk_trace_ray");
		const int r = rtk_host_trace_ray(scene, ray, hit, nullptr);
		if (r < 0) { t_fatal = true; return per_ray_failure("rtk_trace_ray"); }       // a blob that is not a tree stays one
		t_per_ray_failures = 0;
		return r == 1;
	}
	uint8_t m = 0;
	rtk_hit h;
	const size_t r = rtk_trace_rays(scene, ray, 1, &h, &m);
	if (r == (size_t)-1) return per_ray_failure("rtk_trace_ray");
	t_per_ray_failures = 0;
	if (m) *hit = h;
	return m != 0;
}

// reference rtk.h:117,130 (stub at rtk.c:579-582): the closest hit that the filter accepts; see
// rtk_trace_rays_filter for the order in which candidates are offered.
extern "C" bool rtk_trace_ray_filter(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, rtk_filter_fn *filter, void *filter_user)
{
	if (!scene || !ray || !hit) { rtk_set_error("rtk_trace_ray_filter: NULL argument"); return false; }
	if (!filter) return rtk_trace_ray(scene, ray, hit);
	if (!per_ray_on_gpu()) {
		if (injected_failure("rtk_trace_ray_filter")) return per_ray_failure("rtk_trace_ray_filter");
		const int r = rtk_host_trace_ray_filter(scene, ray, hit, filter, filter_user);
		if (r < 0) { t_fatal = true; return per_ray_failure("rtk_trace_ray_filter"); }
		t_per_ray_failures = 0;
		return r == 1;
	}
	uint8_t m = 0;
	rtk_hit h;
	const size_t r = rtk_trace_rays_filter(scene, ray, 1, &h, &m, filter, filter_user);
	if (r == (size_t)-1) return per_ray_failure("rtk_trace_ray_filter");
	t_per_ray_failures = 0;
	if (m) *hit = h;
	return m != 0;
}
