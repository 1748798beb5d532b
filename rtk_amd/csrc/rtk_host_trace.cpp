// rtk_host_trace.cpp -- rtk_trace_ray / rtk_trace_ray_filter served on the calling thread ("rtk_trace_ray itself
// is served by the CPU backend").
//
// A synchronous call for ONE ray cannot drive a GPU: a launch + ticket round trip is 6.7 us on this machine before any work
// (profiles/r04_launch_latency_probe.log), the reference's call is under a microsecond on a core (rtk.c:543-577). So the two
// per-ray symbols of rtk.h walk the caller's blob where it lies, on the calling thread: no residency, no stream, no staging.
// This is the per-ray symbols' implementation, not a fallback: nothing else ever routes here -- every batch entry point
// (rtk_trace_rays*, rtk_dev_*, rtk_mgpu_*) runs the HIP kernels or fails -- and RTK_AMD_PER_RAY=gpu sends the per-ray
// symbols through the one-ray kernel again (rtk_capi.hip).
//
// What it computes is what the GPU's exact path computes on the same blob, bit for bit (tests/test_host_per_ray.py):
//   * node test: (plane - origin) * (1/d) per plane, near / far row picked by the direction's sign BIT, entry = max of the
//     three entry parameters and min_t, exit = min of the three exit parameters and the current hit distance, SSE operand
//     order (a NaN operand loses: rtk.c:458-470);
//   * leaf test in the blob's groups of four slots (padding included): shear-space edge functions in float, ALL FOUR lanes
//     redone in double when any lane has an exact zero (rtk.c:298-336), t = ((u z0 + v z1) + w z2) * (1 / det), no contraction;
//   * canonical ties: among bit-equal t the lowest (mesh, triangle) wins, so the answer does not depend on the order in which
//     children are visited (the reference keeps whichever it met first, rtk.c:371);
//   * the stack fix is moot: children are pushed far to near on an explicit stack.
// Own code, written against the blob format; nothing of oracle/ is included, linked or called.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "rtk.h"
#include "rtk_amd.h"

void rtk_set_error(const char *fmt, ...);

namespace {

struct BlobNode {                 // 128 B (Appendix A)
	float lo_hi[3][2][4];         // [axis][0 = min planes, 1 = max planes][slot]
	uint64_t child[4];            // byte offset, bit 0 = leaf
};
struct BlobTri { uint8_t v[3]; uint8_t local_mesh; uint32_t triangle_index; };

struct Entry { float t; uint32_t pad; uint64_t ptr; };

const int STACK_CAP = 256;              // a tree of depth 64 needs 3 * 64 + 1
const uint32_t STEP_CAP = 1u << 24;     // a blob that is not a tree (a cycle) ends here instead of never

inline bool id_before(uint32_t mesh_a, uint32_t tri_a, uint32_t mesh_b, uint32_t tri_b)
{
	return mesh_a < mesh_b || (mesh_a == mesh_b && tri_a < tri_b);
}

struct Walker {
	const char *blob;
	uint64_t size;
	const rtk_ray *ray;
	// candidates at or before (after_t, after_mesh, after_tri) are not offered (the filter loop's cursor)
	bool has_after;
	float after_t;
	uint32_t after_mesh, after_tri;
	// best so far
	float best_t, best_u, best_v;
	const rtk_vertex *best_verts;
	const BlobTri *best_tri;
	uint32_t best_mesh;
	bool found;
	// shear space (rtk.c:550-566)
	int kx, ky, kz;
	float sox, soy, soz, shx, shy, shz;

	bool leaf(uint64_t ptr);
};

// One leaf: the slots in groups of four, as the blob stores them. Returns false on a malformed leaf.
bool Walker::leaf(uint64_t ptr)
{
	const uint64_t at = ptr & ~(uint64_t)1;
	if (at + 8 > size) return false;
	uint64_t info;
	memcpy(&info, blob + at, 8);
	const size_t num = (size_t)(info & 0x3fu);
	const size_t slots = (num + 3u) & ~(size_t)3u;
	const uint64_t group = info & ~(uint64_t)0x3f;
	if (at + 8 + slots * 8 > size || group + 16 > size) return false;
	const BlobTri *tris = reinterpret_cast<const BlobTri *>(blob + at + 8);
	const uint32_t *mesh_table = reinterpret_cast<const uint32_t *>(tris + slots);
	const rtk_vertex *verts = reinterpret_cast<const rtk_vertex *>(blob + group);
	const uint64_t verts_room = (size - group) / sizeof(rtk_vertex);     // vertex indices are u8: 256 records are enough to be safe
	const bool check_each = verts_room < 256;

	const __m128 o_x = _mm_set1_ps(sox), o_y = _mm_set1_ps(soy), o_z = _mm_set1_ps(soz);
	const __m128 s_x = _mm_set1_ps(shx), s_y = _mm_set1_ps(shy), s_z = _mm_set1_ps(shz);
	const __m128 zero = _mm_setzero_ps();
	const __m128 t_lo = _mm_set1_ps(ray->min_t), t_hi = _mm_set1_ps(ray->max_t);

	for (size_t g = 0; g < slots; g += 4) {
		// corner c of the four triangles as SoA registers, axes already in (kx, ky, kz) order
		__m128 X[3], Y[3], Z[3];
		for (int c = 0; c < 3; c++) {
			if (check_each) for (int l = 0; l < 4; l++) if (tris[g + l].v[c] >= verts_room) return false;
			__m128 r0 = _mm_loadu_ps(&verts[tris[g + 0].v[c]].position.x);
			__m128 r1 = _mm_loadu_ps(&verts[tris[g + 1].v[c]].position.x);
			__m128 r2 = _mm_loadu_ps(&verts[tris[g + 2].v[c]].position.x);
			__m128 r3 = _mm_loadu_ps(&verts[tris[g + 3].v[c]].position.x);
			_MM_TRANSPOSE4_PS(r0, r1, r2, r3);                              // r0 = x of the four, r1 = y, r2 = z
			const __m128 axis[3] = { r0, r1, r2 };
			const __m128 px = _mm_sub_ps(axis[kx], o_x), py = _mm_sub_ps(axis[ky], o_y), pz = _mm_sub_ps(axis[kz], o_z);
			X[c] = _mm_add_ps(px, _mm_mul_ps(s_x, pz));
			Y[c] = _mm_add_ps(py, _mm_mul_ps(s_y, pz));
			Z[c] = _mm_mul_ps(s_z, pz);
		}
		__m128 u = _mm_sub_ps(_mm_mul_ps(X[1], Y[2]), _mm_mul_ps(Y[1], X[2]));
		__m128 v = _mm_sub_ps(_mm_mul_ps(X[2], Y[0]), _mm_mul_ps(Y[2], X[0]));
		__m128 w = _mm_sub_ps(_mm_mul_ps(X[0], Y[1]), _mm_mul_ps(Y[0], X[1]));
		const __m128 zeros = _mm_or_ps(_mm_or_ps(_mm_cmpeq_ps(u, zero), _mm_cmpeq_ps(v, zero)), _mm_cmpeq_ps(w, zero));
		if (_mm_movemask_ps(zeros)) {
			// the whole group again in double precision, rounded once (rtk.c:302-336): low pair, high pair
			__m128 uu[2], vv[2], ww[2];
			for (int h = 0; h < 2; h++) {
				auto half = [h](__m128 r) { return _mm_cvtps_pd(h ? _mm_movehl_ps(r, r) : r); };
				const __m128d x0 = half(X[0]), y0 = half(Y[0]), x1 = half(X[1]), y1 = half(Y[1]), x2 = half(X[2]), y2 = half(Y[2]);
				uu[h] = _mm_cvtpd_ps(_mm_sub_pd(_mm_mul_pd(x1, y2), _mm_mul_pd(y1, x2)));
				vv[h] = _mm_cvtpd_ps(_mm_sub_pd(_mm_mul_pd(x2, y0), _mm_mul_pd(y2, x0)));
				ww[h] = _mm_cvtpd_ps(_mm_sub_pd(_mm_mul_pd(x0, y1), _mm_mul_pd(y0, x1)));
			}
			u = _mm_movelh_ps(uu[0], uu[1]);
			v = _mm_movelh_ps(vv[0], vv[1]);
			w = _mm_movelh_ps(ww[0], ww[1]);
		}
		const __m128 some_neg = _mm_cmplt_ps(_mm_min_ps(_mm_min_ps(u, v), w), zero);
		const __m128 some_pos = _mm_cmpgt_ps(_mm_max_ps(_mm_max_ps(u, v), w), zero);
		const int mixed = _mm_movemask_ps(_mm_and_ps(some_neg, some_pos));
		if (mixed == 0xf) continue;
		const __m128 det = _mm_add_ps(_mm_add_ps(u, v), w);
		const __m128 inv = _mm_div_ps(_mm_set1_ps(1.0f), det);
		__m128 zz = _mm_mul_ps(u, Z[0]);
		zz = _mm_add_ps(zz, _mm_mul_ps(v, Z[1]));
		zz = _mm_add_ps(zz, _mm_mul_ps(w, Z[2]));
		const __m128 t4 = _mm_mul_ps(zz, inv);
		int live = _mm_movemask_ps(_mm_and_ps(_mm_cmpgt_ps(t4, t_lo), _mm_cmplt_ps(t4, t_hi))) & ~mixed;
		if (!live) continue;
		alignas(16) float ts[4], us[4], vs[4];
		_mm_store_ps(ts, t4);
		_mm_store_ps(us, _mm_mul_ps(u, inv));
		_mm_store_ps(vs, _mm_mul_ps(v, inv));
		for (; live; live &= live - 1) {
			const int l = __builtin_ctz((unsigned)live);
			const float t = ts[l];
			const BlobTri *tri = &tris[g + l];
			if (!(t <= best_t)) continue;                                   // cheap reject before the id is looked up
			if (at + 8 + slots * 8 + 4 * ((uint64_t)tri->local_mesh + 1) > size) return false;     // (the leaf-local mesh table ends inside the blob)
			const uint32_t mesh = mesh_table[tri->local_mesh];
			if (t == best_t && !(found && id_before(mesh, tri->triangle_index, best_mesh, best_tri->triangle_index))) continue;
			if (has_after && !(t > after_t || (t == after_t && id_before(after_mesh, after_tri, mesh, tri->triangle_index)))) continue;
			best_t = t; best_u = us[l]; best_v = vs[l];
			best_verts = verts; best_tri = tri; best_mesh = mesh;
			found = true;
		}
	}
	return true;
}

}  // namespace

// 1 = hit (*hit written), 0 = miss (*hit untouched, rtk.c:571-576), -1 = the blob is not a traversable tree (error text set).
// `after`: only candidates behind (after->t, after->mesh_index, after->triangle_index) in (t, mesh, triangle) order.
int rtk_host_trace_ray(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, const rtk_hit *after)
{
	Walker w;
	w.blob = reinterpret_cast<const char *>(scene);
	w.size = scene->size_in_bytes;
	if (w.size < 256) { rtk_set_error("rtk_trace_ray: scene blob of %llu bytes has no root node", (unsigned long long)w.size); return -1; }
	w.ray = ray;
	w.has_after = after != nullptr;
	w.after_t = after ? after->t : 0.0f;
	w.after_mesh = after ? after->mesh_index : 0u;
	w.after_tri = after ? after->triangle_index : 0u;
	w.best_t = ray->max_t;
	w.best_u = w.best_v = 0.0f;
	w.best_verts = nullptr; w.best_tri = nullptr; w.best_mesh = 0u;
	w.found = false;

	// dominant axis: the FIRST axis whose |d| equals the largest (x before y before z); max with SSE operand order
	const float *d = ray->direction.v, *o = ray->origin.v;
	uint32_t bits[3];
	memcpy(bits, d, 12);
	float a[3];
	for (int k = 0; k < 3; k++) { const uint32_t m = bits[k] & 0x7fffffffu; memcpy(&a[k], &m, 4); }
	float big = a[0] > a[1] ? a[0] : a[1];
	big = big > a[2] ? big : a[2];
	w.kz = a[0] == big ? 0 : a[1] == big ? 1 : 2;
	w.kx = (w.kz + 1) % 3;
	w.ky = (w.kz + 2) % 3;
	w.shx = -d[w.kx] / d[w.kz];
	w.shy = -d[w.ky] / d[w.kz];
	w.shz = 1.0f / d[w.kz];
	w.sox = o[w.kx]; w.soy = o[w.ky]; w.soz = o[w.kz];

	const int sx = (int)(bits[0] >> 31), sy = (int)(bits[1] >> 31), sz = (int)(bits[2] >> 31);      // sign BITS: -0.0 is negative (rtk.c:152-154)
	const __m128 rd = _mm_div_ps(_mm_set1_ps(1.0f), _mm_setr_ps(d[0], d[1], d[2], 1.0f));
	const __m128 ox = _mm_set1_ps(o[0]), oy = _mm_set1_ps(o[1]), oz = _mm_set1_ps(o[2]);
	const __m128 rdx = _mm_shuffle_ps(rd, rd, 0x00), rdy = _mm_shuffle_ps(rd, rd, 0x55), rdz = _mm_shuffle_ps(rd, rd, 0xaa);
	const __m128 t_min = _mm_set1_ps(ray->min_t);

	Entry stack[STACK_CAP];
	int depth = 0;
	stack[depth++] = Entry{ -RTK_INF, 0u, 128u };                            // the root is the node at byte 128 (rtk.c:569)
	uint32_t steps = 0;
	while (depth) {
		const Entry e = stack[--depth];
		if (e.t > w.best_t) continue;                                        // behind the hit; equal distances are still looked at (ties)
		if (++steps > STEP_CAP) { rtk_set_error("rtk_trace_ray: more than 2^24 traversal steps: the scene blob is not a tree"); return -1; }
		if (e.ptr & 1u) {
			if (!w.leaf(e.ptr)) { rtk_set_error("rtk_trace_ray: a leaf of the scene blob points outside its %llu bytes", (unsigned long long)w.size); return -1; }
			continue;
		}
		if (e.ptr + sizeof(BlobNode) > w.size) { rtk_set_error("rtk_trace_ray: a node of the scene blob lies outside its %llu bytes", (unsigned long long)w.size); return -1; }
		const BlobNode *n = reinterpret_cast<const BlobNode *>(w.blob + e.ptr);
		const __m128 nx = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(n->lo_hi[0][sx]), ox), rdx), fx = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(n->lo_hi[0][sx ^ 1]), ox), rdx);
		const __m128 ny = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(n->lo_hi[1][sy]), oy), rdy), fy = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(n->lo_hi[1][sy ^ 1]), oy), rdy);
		const __m128 nz = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(n->lo_hi[2][sz]), oz), rdz), fz = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(n->lo_hi[2][sz ^ 1]), oz), rdz);
		const __m128 enter = _mm_max_ps(_mm_max_ps(nx, ny), _mm_max_ps(nz, t_min));
		const __m128 leave = _mm_min_ps(_mm_min_ps(fx, fy), _mm_min_ps(fz, _mm_set1_ps(w.best_t)));
		int in = _mm_movemask_ps(_mm_cmple_ps(enter, leave));
		if (!in) continue;
		if (depth + 4 > STACK_CAP) { rtk_set_error("rtk_trace_ray: traversal stack of %d entries exhausted: the scene blob is deeper than a valid tree", STACK_CAP); return -1; }
		alignas(16) float te[4];
		_mm_store_ps(te, enter);
		// pushed far to near: insertion into the (at most four) new entries, descending by entry distance
		const int base = depth;
		for (; in; in &= in - 1) {
			const int k = __builtin_ctz((unsigned)in);
			int at = depth++;
			while (at > base && stack[at - 1].t < te[k]) { stack[at] = stack[at - 1]; at--; }
			stack[at] = Entry{ te[k], 0u, n->child[k] };
		}
	}
	if (!(w.best_t < ray->max_t) || !w.found) return 0;
	hit->t = w.best_t; hit->u = w.best_u; hit->v = w.best_v;
	hit->vertex[0] = w.best_verts[w.best_tri->v[0]];
	hit->vertex[1] = w.best_verts[w.best_tri->v[1]];
	hit->vertex[2] = w.best_verts[w.best_tri->v[2]];
	hit->mesh_index = w.best_mesh;
	hit->triangle_index = w.best_tri->triangle_index;
	return 1;
}

// rtk.h:130 on the host: the candidates of the ray in increasing (t, mesh, triangle) order, every one of them (equal t
// included), until the callback accepts one -- the order rtk_trace_rays_filter offers them in (include/rtk_amd.h). One walk per
// rejected candidate.
int rtk_host_trace_ray_filter(const rtk_scene *scene, const rtk_ray *ray, rtk_hit *hit, rtk_filter_fn *filter, void *user)
{
	rtk_hit cand, cursor;
	bool have_cursor = false;
	for (;;) {
		const int r = rtk_host_trace_ray(scene, ray, &cand, have_cursor ? &cursor : nullptr);
		if (r <= 0) return r;
		if (filter(user, ray, &cand)) { *hit = cand; return 1; }
		cursor = cand;
		have_cursor = true;
	}
}
