// rtk_capi.hip -- the C-ABI of librtk_amd.so: the nine rtk.h entry points plus the
// additive batch/device surface of rtk_amd.h. Thin: argument checks, residency cache,
// HIP plumbing. No intersection arithmetic lives here and there is no CPU fallback:
// every trace call runs the HIP kernels of rtk_trace.hip or fails with an error.
#include "rtk_dev.h"
#include "rtk_layout_check.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
	if (rtk_blob_to_host_bvh(scene, &h) != RTK_AMD_OK) return nullptr;
	return rtk_dev_scene_from_host_bvh(h);
}

void rtk_export_forget(const rtk_dev_scene *ds);   // rtk_build.hip

extern "C" void rtk_dev_scene_free(rtk_dev_scene *ds)
{
	if (!ds) return;
	rtk_export_forget(ds);
	for (void *p : ds->allocs) (void)hipFree(p);
	if (ds->d_counter) (void)hipFree(ds->d_counter);
	if (ds->d_spill) (void)hipFree(ds->d_spill);
	if (ds->d_sort) (void)hipFree(ds->d_sort);
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

extern "C" int rtk_dev_expand_hits(const rtk_dev_scene *ds, const rtk_hit_record *d_records, size_t n,
	rtk_hit *d_hits, uint8_t *d_mask, void *stream)
{
	return rtk_launch_expand(ds, d_records, n, d_hits, d_mask, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------- residency cache

static std::mutex g_cache_mutex;
static std::unordered_map<const rtk_scene *, rtk_dev_scene *> g_cache;

static rtk_dev_scene *resident(const rtk_scene *scene)
{
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	auto it = g_cache.find(scene);
	if (it != g_cache.end()) return it->second;
	rtk_dev_scene *ds = rtk_dev_scene_upload(scene);
	if (ds) g_cache[scene] = ds;
	return ds;
}

void rtk_cache_adopt(const rtk_scene *scene, rtk_dev_scene *ds)
{
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	auto it = g_cache.find(scene);
	if (it != g_cache.end()) rtk_dev_scene_free(it->second);
	g_cache[scene] = ds;
}

extern "C" void rtk_amd_forget_scene(const rtk_scene *scene)
{
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	auto it = g_cache.find(scene);
	if (it == g_cache.end()) return;
	rtk_dev_scene_free(it->second);
	g_cache.erase(it);
}

extern "C" size_t rtk_trace_rays(const rtk_scene *scene, const rtk_ray *rays, size_t n, rtk_hit *hits, uint8_t *hit_mask)
{
	if (!scene || (!rays && n)) { rtk_set_error("rtk_trace_rays: NULL argument"); return (size_t)-1; }
	if (n == 0) return 0;
	rtk_dev_scene *ds = resident(scene);
	if (!ds) return (size_t)-1;

	rtk_ray *d_rays = nullptr;
	rtk_hit_record *d_rec = nullptr;
	rtk_hit *d_hits = nullptr;
	uint8_t *d_mask = nullptr;
	size_t result = (size_t)-1;
	std::vector<rtk_hit> h_hits;
	std::vector<uint8_t> h_mask(n);
	do {
		if (hipMalloc(&d_rays, n * sizeof(rtk_ray)) != hipSuccess || hipMalloc(&d_rec, n * sizeof(rtk_hit_record)) != hipSuccess ||
			hipMalloc(&d_mask, n) != hipSuccess || (hits && hipMalloc(&d_hits, n * sizeof(rtk_hit)) != hipSuccess)) {
			rtk_set_error("rtk_trace_rays: device allocation failed"); break;
		}
		if (hipMemcpy(d_rays, rays, n * sizeof(rtk_ray), hipMemcpyHostToDevice) != hipSuccess) { rtk_set_error("rtk_trace_rays: H2D copy failed"); break; }
		if (rtk_launch_trace(ds, d_rays, n, d_rec, nullptr, nullptr, nullptr, false, nullptr) != RTK_AMD_OK) break;
		if (rtk_launch_expand(ds, d_rec, n, d_hits, d_mask, nullptr) != RTK_AMD_OK) break;
		if (hipMemcpy(h_mask.data(), d_mask, n, hipMemcpyDeviceToHost) != hipSuccess) { rtk_set_error("rtk_trace_rays: D2H copy failed: %s", hipGetErrorString(hipGetLastError())); break; }
		if (hits) {
			h_hits.resize(n);
			if (hipMemcpy(h_hits.data(), d_hits, n * sizeof(rtk_hit), hipMemcpyDeviceToHost) != hipSuccess) { rtk_set_error("rtk_trace_rays: D2H copy failed"); break; }
		}
		size_t count = 0;
		for (size_t i = 0; i < n; i++) {
			if (h_mask[i]) { count++; if (hits) hits[i] = h_hits[i]; }   // misses stay untouched (rtk.c:571-576)
			if (hit_mask) hit_mask[i] = h_mask[i];
		}
		result = count;
	} while (0);
	if (d_rays) (void)hipFree(d_rays);
	if (d_rec) (void)hipFree(d_rec);
	if (d_hits) (void)hipFree(d_hits);
	if (d_mask) (void)hipFree(d_mask);
	return result;
}

// ---------------------------------------------------------------------------- rtk.h: trace

// reference rtk.h:129 / rtk.c:543-577 -- a batch of one on the GPU. Correct and
// re-entrant, but a launch per ray: throughput callers use rtk_trace_rays.
extern "C" bool rtk_trace_ray(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit)
{
	uint8_t m = 0;
	rtk_hit h;
	if (!ray || !hit) return false;
	const size_t r = rtk_trace_rays(scene, ray, 1, &h, &m);
	if (r == (size_t)-1) {
		fprintf(stderr, "rtk_trace_ray: %s\n", g_error);
		abort();   // never answer "miss" because the GPU path is unavailable
	}
	if (m) *hit = h;
	return m != 0;
}

// reference rtk.h:117,130 (stub at rtk.c:579-582). Semantics defined here: the closest
// hit that the filter accepts. Candidates are offered in increasing t; a rejected
// candidate moves the open interval past its t.
extern "C" bool rtk_trace_ray_filter(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, rtk_filter_fn *filter, void *filter_user)
{
	if (!filter) return rtk_trace_ray(scene, ray, hit);
	rtk_ray r = *ray;
	for (int guard = 0; guard < (1 << 20); guard++) {
		rtk_hit h;
		if (!rtk_trace_ray(scene, &r, &h)) return false;
		if (filter(filter_user, ray, &h)) { *hit = h; return true; }
		r.min_t = h.t;
	}
	return false;
}
