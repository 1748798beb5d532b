// rtk_mgpu.hip -- several GPUs of one node behind the C ABI, one process.
//
// Rays are independent and the BVH is read-only, so the path shards by ray range and nothing else:
//   * the scene is replicated: the deterministic device build (or the blob upload) runs on every GPU,
//     which costs no traffic at all and gives byte-identical trees (rtk_dev_scene_validate's hash);
//   * shard r of R owns the contiguous range rtk_amd_shard_range(n, r, R) of the batch, so the gather
//     of the results is a concatenation at known offsets;
//   * the only exchange is that gather. Every GPU's records travel straight to their final place
//     (the caller's host array, or one root GPU's buffer) on the sender's own copy stream: on the xGMI
//     mesh each sender has its own link into the root, so nothing is relayed and no ring is formed.
//     A shard is traced in pieces and each piece's records leave while the next piece is being traced
//     (a 2^24-ray shard produces 256 MB of records; one link direction carries ~77 GB/s).
//   * a gather onto ONE root is bound by that root's links (a shard's 268 MB need 3.5 ms on its one link into the root and
//     are traced in 1.0 ms: eight GPUs gathered onto one deliver ~2x one GPU). The STRIPED form
//     (rtk_mgpu_trace_rays_device_striped) is the answer: GPU j receives stripe j of EVERY shard, so every GPU sends
//     1/R of each piece over each of its links and no link carries more than 1/R of a shard; the result lives striped
//     across the GPUs (rtk_mgpu_striped_segment says where), the same layout as rtk_amd/shard.py's exchange_striped_start.
// No collective library is involved: peer copies are all these exchanges need; the per-process
// torch.distributed/RCCL form of the same partitioning lives in rtk_amd/shard.py for bench.py --gpus N.
#include "rtk_dev.h"

#include <string.h>

#include <string>
#include <thread>
#include <vector>

namespace {

const size_t MGPU_PIECE = (size_t)1 << 21;      // rays per piece: 32 MB of records per copy

struct DeviceSlot {
	int device = 0;
	rtk_dev_scene *scene = nullptr;
	hipStream_t trace_stream = nullptr, copy_stream = nullptr;
	std::vector<hipEvent_t> events;            // one per piece in flight
	rtk_ray *d_rays = nullptr;                 // staging for host-pointer calls
	rtk_hit_record *d_rec = nullptr;
	size_t cap = 0;
};

} // namespace

struct rtk_mgpu {
	std::vector<DeviceSlot> slots;
};

extern "C" void rtk_amd_shard_range(size_t n, int rank, int num_shards, size_t *first, size_t *count)
{
	// [floor(r n / R), floor((r + 1) n / R)): the same rule as rtk_amd/shard.py (the per-process RCCL form)
	size_t f = 0, c = 0;
	if (num_shards > 0 && rank >= 0 && rank < num_shards) {
		const unsigned __int128 N = n;
		f = (size_t)(N * (unsigned)rank / (unsigned)num_shards);
		c = (size_t)(N * ((unsigned)rank + 1u) / (unsigned)num_shards) - f;
	}
	if (first) *first = f;
	if (count) *count = c;
}

extern "C" rtk_mgpu *rtk_mgpu_create(const int *devices, int num_devices)
{
	int have = 0;
	if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { rtk_set_error("rtk_mgpu_create: no HIP device"); return nullptr; }
	if (num_devices <= 0) { num_devices = have; devices = nullptr; }
	if (num_devices > RTK_MAX_DEVICES) { rtk_set_error("rtk_mgpu_create: too many devices"); return nullptr; }
	int before = 0;
	(void)hipGetDevice(&before);
	rtk_mgpu *m = new rtk_mgpu();
	m->slots.resize(num_devices);
	bool ok = true;
	for (int i = 0; i < num_devices && ok; i++) {
		DeviceSlot &s = m->slots[i];
		s.device = devices ? devices[i] : i;
		if (s.device < 0 || s.device >= have) { rtk_set_error("rtk_mgpu_create: device %d does not exist", s.device); ok = false; break; }
		ok = hipSetDevice(s.device) == hipSuccess && hipStreamCreateWithFlags(&s.trace_stream, hipStreamNonBlocking) == hipSuccess &&
			hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking) == hipSuccess;
		if (!ok) rtk_set_error("rtk_mgpu_create: device %d: %s", s.device, hipGetErrorString(hipGetLastError()));
	}
	// every GPU may write into every other one's memory (the gather target)
	for (int i = 0; i < num_devices && ok; i++) {
		(void)hipSetDevice(m->slots[i].device);
		for (int j = 0; j < num_devices; j++) {
			if (m->slots[j].device == m->slots[i].device) continue;
			const hipError_t e = hipDeviceEnablePeerAccess(m->slots[j].device, 0);
			if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // copies then go through the host; still correct
		}
	}
	(void)hipSetDevice(before);
	if (!ok) { rtk_mgpu_destroy(m); return nullptr; }
	return m;
}

extern "C" void rtk_mgpu_destroy(rtk_mgpu *m)
{
	if (!m) return;
	int before = 0;
	(void)hipGetDevice(&before);
	for (DeviceSlot &s : m->slots) {
		(void)hipSetDevice(s.device);
		if (s.scene) rtk_dev_scene_free(s.scene);
		for (hipEvent_t e : s.events) (void)hipEventDestroy(e);
		if (s.d_rays) (void)hipFree(s.d_rays);
		if (s.d_rec) (void)hipFree(s.d_rec);
		if (s.trace_stream) (void)hipStreamDestroy(s.trace_stream);
		if (s.copy_stream) (void)hipStreamDestroy(s.copy_stream);
	}
	(void)hipSetDevice(before);
	delete m;
}

extern "C" int rtk_mgpu_num_devices(const rtk_mgpu *m) { return m ? (int)m->slots.size() : 0; }

extern "C" const rtk_dev_scene *rtk_mgpu_scene(const rtk_mgpu *m, int index)
{
	return (m && index >= 0 && index < (int)m->slots.size()) ? m->slots[index].scene : nullptr;
}

// One host thread per device slot runs `work(slot index)`; the first failure's code and message (error strings are per
// thread) come back to the caller. Slots that share a physical device (virtual shards) are run by one thread, in order.
template <typename F>
static int for_each_slot_in_parallel(rtk_mgpu *m, F work)
{
	const size_t R = m->slots.size();
	std::vector<int> rcs(R, RTK_AMD_OK);
	std::vector<std::string> msgs(R);
	std::vector<std::thread> threads;
	std::vector<bool> taken(R, false);
	for (size_t i = 0; i < R; i++) {
		if (taken[i]) continue;
		std::vector<size_t> mine;
		for (size_t j = i; j < R; j++) if (!taken[j] && m->slots[j].device == m->slots[i].device) { mine.push_back(j); taken[j] = true; }
		threads.emplace_back([&, mine]() {
			for (size_t j : mine) {
				if (hipSetDevice(m->slots[j].device) != hipSuccess) { rcs[j] = RTK_AMD_ERR_NO_DEVICE; msgs[j] = "rtk_mgpu: hipSetDevice failed"; (void)hipGetLastError(); continue; }
				rcs[j] = work(j);
				if (rcs[j] != RTK_AMD_OK) msgs[j] = rtk_amd_last_error();
			}
		});
	}
	for (std::thread &t : threads) t.join();
	for (size_t j = 0; j < R; j++) if (rcs[j] != RTK_AMD_OK) { rtk_set_error("device slot %zu: %s", j, msgs[j].c_str()); return rcs[j]; }
	return RTK_AMD_OK;
}

// the scene on every GPU of the context: each device builds (or uploads) its own replica on a host thread of its own, so
// eight replicas take the time of one
static int replicate(rtk_mgpu *m, const rtk_scene_desc *desc, const rtk_scene *blob)
{
	if (!m) { rtk_set_error("rtk_mgpu: NULL context"); return RTK_AMD_ERR_BAD_ARG; }
	return for_each_slot_in_parallel(m, [&](size_t j) -> int {
		DeviceSlot &s = m->slots[j];
		if (s.scene) rtk_dev_scene_free(s.scene);
		s.scene = desc ? rtk_dev_scene_build(desc) : rtk_dev_scene_upload(blob);
		return s.scene ? RTK_AMD_OK : RTK_AMD_ERR_HIP;
	});
}

extern "C" int rtk_mgpu_build(rtk_mgpu *m, const rtk_scene_desc *desc)
{
	if (!desc) { rtk_set_error("rtk_mgpu_build: NULL description"); return RTK_AMD_ERR_BAD_ARG; }
	return replicate(m, desc, nullptr);
}

extern "C" int rtk_mgpu_upload(rtk_mgpu *m, const rtk_scene *scene)
{
	if (!scene) { rtk_set_error("rtk_mgpu_upload: NULL scene"); return RTK_AMD_ERR_BAD_ARG; }
	return replicate(m, nullptr, scene);
}

// Where stripe `stripe` of shard `shard` lies: records [*first, *first + *count) of the buffer of GPU `stripe` (stripes of a
// shard follow rtk_amd_shard_range(counts[shard], stripe, R); a GPU's buffer holds its stripe of shard 0, then of shard 1, ...).
extern "C" void rtk_mgpu_striped_segment(const size_t *counts, int num_shards, int shard, int stripe, size_t *first, size_t *count)
{
	size_t at = 0, c = 0;
	if (counts && num_shards > 0 && shard >= 0 && shard < num_shards && stripe >= 0 && stripe < num_shards) {
		for (int r = 0; r < shard; r++) {
			size_t f, k;
			rtk_amd_shard_range(counts[r], stripe, num_shards, &f, &k);
			at += k;
		}
		size_t f;
		rtk_amd_shard_range(counts[shard], stripe, num_shards, &f, &c);
	}
	if (first) *first = at;
	if (count) *count = c;
}

// One copy of `m` records from this slot's d_src to `dst`, which is host memory (dst_device < 0) or memory of device dst_device
static int copy_records(DeviceSlot &s, rtk_hit_record *dst, int dst_device, const rtk_hit_record *d_src, size_t m)
{
	if (m == 0 || dst == d_src) return RTK_AMD_OK;
	if (dst_device < 0) RTK_HIP_CHECK(hipMemcpyAsync(dst, d_src, m * sizeof(rtk_hit_record), hipMemcpyDeviceToHost, s.copy_stream), RTK_AMD_ERR_HIP);
	else if (dst_device == s.device) RTK_HIP_CHECK(hipMemcpyAsync(dst, d_src, m * sizeof(rtk_hit_record), hipMemcpyDeviceToDevice, s.copy_stream), RTK_AMD_ERR_HIP);
	else RTK_HIP_CHECK(hipMemcpyPeerAsync(dst, dst_device, d_src, s.device, m * sizeof(rtk_hit_record), s.copy_stream), RTK_AMD_ERR_HIP);
	return RTK_AMD_OK;
}

// Trace a shard (rays already on its GPU at d_rays, `count` of them) in pieces; as soon as a piece is done its records
// [at, at + m) of the shard are sent on their way by send(at, m), enqueued on the slot's copy stream.
template <typename Send>
static int trace_shard(DeviceSlot &s, const rtk_ray *d_rays, size_t count, rtk_hit_record *d_rec, const rtk_trace_opts *opts, Send send)
{
	const size_t pieces = (count + MGPU_PIECE - 1) / MGPU_PIECE;
	while (s.events.size() < pieces) {
		hipEvent_t e;
		RTK_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming), RTK_AMD_ERR_HIP);
		s.events.push_back(e);
	}
	// an image-shaped shard stays image shaped piece by piece (bands of whole 8-row tile rows), so that the packet
	// kernel is used; anything else is traced as plain batches
	rtk_trace_opts band;
	memset(&band, 0, sizeof(band));
	const bool image = opts && opts->struct_size >= 16 && opts->image_width && opts->image_height &&
		(size_t)opts->image_width * opts->image_height == count && opts->image_width % 8u == 0 && opts->image_height % 8u == 0 &&
		MGPU_PIECE % ((size_t)opts->image_width * 8u) == 0;
	if (opts) memcpy(&band, opts, opts->struct_size < sizeof(band) ? opts->struct_size : sizeof(band));
	for (size_t k = 0; k < pieces; k++) {
		const size_t at = k * MGPU_PIECE, m = count - at < MGPU_PIECE ? count - at : MGPU_PIECE;
		if (image) band.image_height = (uint32_t)(m / opts->image_width);
		else { band.image_width = 0; band.image_height = 0; }
		const rtk_trace_opts *o = opts ? &band : nullptr;
		const int rc = rtk_launch_trace(s.scene, d_rays + at, m, d_rec + at, nullptr, o, s.trace_stream, false, nullptr);
		if (rc != RTK_AMD_OK) return rc;
		RTK_HIP_CHECK(hipEventRecord(s.events[k], s.trace_stream), RTK_AMD_ERR_HIP);
		RTK_HIP_CHECK(hipStreamWaitEvent(s.copy_stream, s.events[k], 0), RTK_AMD_ERR_HIP);
		const int crc = send(at, m);
		if (crc != RTK_AMD_OK) return crc;
	}
	return RTK_AMD_OK;
}

static int finish_all(rtk_mgpu *m)
{
	int rc = RTK_AMD_OK;
	for (DeviceSlot &s : m->slots) {
		if (hipSetDevice(s.device) != hipSuccess || hipStreamSynchronize(s.copy_stream) != hipSuccess) { rtk_set_error("rtk_mgpu: device %d: %s", s.device, hipGetErrorString(hipGetLastError())); rc = RTK_AMD_ERR_HIP; continue; }
		const int st = rtk_trace_status(s.scene, s.trace_stream);
		if (st != RTK_AMD_OK) rc = st;
	}
	return rc;
}

extern "C" int rtk_mgpu_trace_rays(rtk_mgpu *m, const rtk_ray *rays, size_t n, rtk_hit_record *records, const rtk_trace_opts *opts)
{
	if (!m || (!rays && n) || (!records && n)) { rtk_set_error("rtk_mgpu_trace_rays: NULL argument"); return RTK_AMD_ERR_BAD_ARG; }
	if (n == 0) return RTK_AMD_OK;
	const int R = (int)m->slots.size();
	for (DeviceSlot &s : m->slots) if (!s.scene) { rtk_set_error("rtk_mgpu_trace_rays: no scene (call rtk_mgpu_build or rtk_mgpu_upload first)"); return RTK_AMD_ERR_BAD_ARG; }
	int before = 0;
	(void)hipGetDevice(&before);
	// an image-shaped batch (rows of `image_width` rays) stays image shaped on a shard that is a whole number of 8-row
	// tile rows -- e.g. eight frames on eight GPUs --, so that every GPU runs the packet kernels, not only a lone one
	const bool image = opts && opts->struct_size >= 16 && opts->image_width && opts->image_height &&
		(size_t)opts->image_width * opts->image_height == n && opts->image_width % 8u == 0;
	// one host thread per device: its shard's rays go up (pageable host memory: the copy is the host's work), are traced in
	// pieces, and the records of each piece come back while the next piece is traced
	int rc = for_each_slot_in_parallel(m, [&](size_t j) -> int {
		DeviceSlot &s = m->slots[j];
		size_t first, count;
		rtk_amd_shard_range(n, (int)j, R, &first, &count);
		if (count == 0) return RTK_AMD_OK;
		if (s.cap < count) {
			if (s.d_rays) (void)hipFree(s.d_rays);
			if (s.d_rec) (void)hipFree(s.d_rec);
			s.d_rays = nullptr; s.d_rec = nullptr; s.cap = 0;
			if (hipMalloc(&s.d_rays, count * sizeof(rtk_ray)) != hipSuccess || hipMalloc(&s.d_rec, count * sizeof(rtk_hit_record)) != hipSuccess) {
				rtk_set_error("rtk_mgpu_trace_rays: out of device memory on device %d", s.device);
				return RTK_AMD_ERR_OOM;
			}
			s.cap = count;
		}
		if (hipMemcpyAsync(s.d_rays, rays + first, count * sizeof(rtk_ray), hipMemcpyHostToDevice, s.trace_stream) != hipSuccess) { rtk_set_error("rtk_mgpu_trace_rays: H2D copy failed"); return RTK_AMD_ERR_HIP; }
		rtk_trace_opts shard_opts;
		const rtk_trace_opts *so = nullptr;
		if (image && first % ((size_t)opts->image_width * 8u) == 0 && count % ((size_t)opts->image_width * 8u) == 0) {
			memset(&shard_opts, 0, sizeof(shard_opts));
			memcpy(&shard_opts, opts, opts->struct_size < sizeof(shard_opts) ? opts->struct_size : sizeof(shard_opts));
			shard_opts.image_height = (uint32_t)(count / opts->image_width);
			so = &shard_opts;
		} else if (opts && !image) so = opts;
		return trace_shard(s, s.d_rays, count, s.d_rec, so, [&](size_t at, size_t m) { return copy_records(s, records + first + at, -1, s.d_rec + at, m); });
	});
	const int frc = finish_all(m);
	(void)hipSetDevice(before);
	return rc != RTK_AMD_OK ? rc : frc;
}

extern "C" int rtk_mgpu_trace_rays_device(rtk_mgpu *m, const rtk_ray *const *d_rays, const size_t *counts, rtk_hit_record *const *d_records,
	rtk_hit_record *d_gathered, int root_index, const rtk_trace_opts *opts)
{
	if (!m || !d_rays || !counts || !d_records) { rtk_set_error("rtk_mgpu_trace_rays_device: NULL argument"); return RTK_AMD_ERR_BAD_ARG; }
	const int R = (int)m->slots.size();
	if (d_gathered && (root_index < 0 || root_index >= R)) { rtk_set_error("rtk_mgpu_trace_rays_device: bad root"); return RTK_AMD_ERR_BAD_ARG; }
	int before = 0, rc = RTK_AMD_OK;
	(void)hipGetDevice(&before);
	size_t offset = 0;
	for (int r = 0; r < R && rc == RTK_AMD_OK; r++) {
		DeviceSlot &s = m->slots[r];
		if (!s.scene) { rtk_set_error("rtk_mgpu_trace_rays_device: no scene"); rc = RTK_AMD_ERR_BAD_ARG; break; }
		if (counts[r] && hipSetDevice(s.device) != hipSuccess) { rc = RTK_AMD_ERR_NO_DEVICE; break; }
		if (counts[r]) {
			rtk_hit_record *const rec = d_records[r];
			rtk_hit_record *const dst = d_gathered ? d_gathered + offset : nullptr;
			const int dst_device = m->slots[root_index >= 0 && root_index < R ? root_index : 0].device;
			rc = trace_shard(s, d_rays[r], counts[r], rec, opts, [&](size_t at, size_t k) { return dst ? copy_records(s, dst + at, dst_device, rec + at, k) : RTK_AMD_OK; });
		}
		offset += counts[r];
	}
	const int frc = finish_all(m);
	(void)hipSetDevice(before);
	return rc != RTK_AMD_OK ? rc : frc;
}

// The exchange that no single link funnels: GPU j of the context ends up with stripe j of EVERY shard (d_striped[j], a buffer
// on GPU j of at least the sum of rtk_mgpu_striped_segment(counts, R, r, j) records), the pieces of a shard leaving in R
// slices over R different links while the rest of the shard is still being traced. d_records[r] keeps the whole shard.
extern "C" int rtk_mgpu_trace_rays_device_striped(rtk_mgpu *m, const rtk_ray *const *d_rays, const size_t *counts, rtk_hit_record *const *d_records,
	rtk_hit_record *const *d_striped, const rtk_trace_opts *opts)
{
	if (!m || !d_rays || !counts || !d_records || !d_striped) { rtk_set_error("rtk_mgpu_trace_rays_device_striped: NULL argument"); return RTK_AMD_ERR_BAD_ARG; }
	const int R = (int)m->slots.size();
	int before = 0, rc = RTK_AMD_OK;
	(void)hipGetDevice(&before);
	for (int r = 0; r < R && rc == RTK_AMD_OK; r++) {
		DeviceSlot &s = m->slots[r];
		if (!s.scene) { rtk_set_error("rtk_mgpu_trace_rays_device_striped: no scene"); rc = RTK_AMD_ERR_BAD_ARG; break; }
		if (counts[r] == 0) continue;
		if (hipSetDevice(s.device) != hipSuccess) { rc = RTK_AMD_ERR_NO_DEVICE; break; }
		// where each stripe of this shard begins inside the shard, and inside its destination GPU's buffer
		std::vector<size_t> sb(R + 1), seg(R);
		for (int j = 0; j < R; j++) {
			size_t f, c, at, len;
			rtk_amd_shard_range(counts[r], j, R, &f, &c);
			rtk_mgpu_striped_segment(counts, R, r, j, &at, &len);
			sb[j] = f; seg[j] = at;
		}
		sb[R] = counts[r];
		rtk_hit_record *const rec = d_records[r];
		rc = trace_shard(s, d_rays[r], counts[r], rec, opts, [&](size_t at, size_t k) -> int {
			// the part of piece [at, at + k) that falls into stripe j goes to GPU j
			for (int j = 0; j < R; j++) {
				const size_t b = at > sb[j] ? at : sb[j], e = at + k < sb[j + 1] ? at + k : sb[j + 1];
				if (e <= b) continue;
				if (!d_striped[j]) { rtk_set_error("rtk_mgpu_trace_rays_device_striped: d_striped[%d] is NULL", j); return RTK_AMD_ERR_BAD_ARG; }
				const int c = copy_records(s, d_striped[j] + seg[j] + (b - sb[j]), m->slots[j].device, rec + b, e - b);
				if (c != RTK_AMD_OK) return c;
			}
			return RTK_AMD_OK;
		});
	}
	const int frc = finish_all(m);
	(void)hipSetDevice(before);
	return rc != RTK_AMD_OK ? rc : frc;
}
