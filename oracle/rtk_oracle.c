/*
 * rtk_oracle.c -- CPU ORACLE for the rtk hot path. TEST INFRASTRUCTURE ONLY
 * (see rtk_oracle.h for who may use it and how it is pinned).
 *
 * Plain scalar C restatement of the reference algorithm; every function cites the
 * reference lines it follows. Where the v0 reference code is defective the intent
 * is followed instead and the SURVEY.md appendix-B id is named.
 *
 * Must be compiled with -ffp-contract=off and without -ffast-math: the sign of the
 * edge functions decides hit/miss, so the float operation order below is normative
 * (SURVEY.md section 0, table).
 */
#include "rtk_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* Blob format (reader side rtk.c:64-86, 181-193)       */
/* ------------------------------------------------------------------------- */

#define ORA_MAX_DEPTH        64   /* rtk.c:5  */
#define ORA_LEAF_MIN_ITEMS   4    /* rtk.c:6  */
#define ORA_LEAF_MAX_ITEMS   63   /* rtk.c:7 says 64 but the count field is 6 bits (rtk.c:188): B7 */
#define ORA_BINS             32   /* rtk.c:587 */
#define ORA_VSET_MAX         256  /* rtk.c:1186 */
#define ORA_ROOT_OFFSET      128  /* rtk.c:569 */

typedef struct {                  /* rtk.c:69-74, 128 bytes */
	float bx[2][4];
	float by[2][4];
	float bz[2][4];
	uint64_t child[4];            /* byte offset; bit 0 set = leaf (rtk.c:64-67) */
} ora_node;

typedef struct {                  /* rtk.c:82-86, 8 bytes */
	uint8_t v[3];
	uint8_t local_mesh;
	uint32_t triangle_index;
} ora_leaf_tri;

/* _mm_min_ps / _mm_max_ps semantics: the SECOND operand is returned when the
 * comparison is false, which includes every NaN case. */
static inline float sse_min(float a, float b) { return a < b ? a : b; }
static inline float sse_max(float a, float b) { return a > b ? a : b; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) & ~(a - 1); }

static void *alloc_aligned(size_t size)
{
	void *p = NULL;
	if (size == 0) size = 128;
	if (posix_memalign(&p, 128, align_up(size, 128)) != 0) return NULL;
	memset(p, 0, align_up(size, 128));
	return p;
}

void ora_free(void *p) { free(p); }

int ora_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Trace                                                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
	const char *blob;
	rtk_ray ray;
	rtk_hit hit;
	bool found;
	int ties;
	uint32_t k[3];        /* kx, ky, kz                                  */
	float so[3];          /* origin permuted to (kx,ky,kz)  rtk.c:564-566 */
	float sh[3];          /* shear constants                rtk.c:561-563 */
	uint32_t sign_mask;
	ora_counters *ctr;
	const ora_filter *filter;   /* optional candidate filter (rtk.h:117 semantics), canonical ties only */
	size_t ray_index;
} ora_trace;

static inline bool id_less(uint32_t mesh_a, uint32_t tri_a, uint32_t mesh_b, uint32_t tri_b);

/* Does the filter let this candidate through? Mirrors rtk_dev_filter of include/rtk_amd.h. */
static bool filter_accepts(const ora_trace *rt, uint32_t mesh, uint32_t tri, float t, const rtk_hit *cand)
{
	const ora_filter *f = rt->filter;
	const size_t i = rt->ray_index;
	if (f->mesh_mask && !(mesh < f->mesh_mask_bits && ((f->mesh_mask[mesh >> 5] >> (mesh & 31u)) & 1u))) return false;
	if (f->ignore_mesh && f->ignore_mesh[i] == mesh && f->ignore_tri[i] == tri) return false;
	if (f->after_t && f->after_mesh[i] != 0xffffffffu) {
		const float at = f->after_t[i];
		if (!(t > at || (t == at && id_less(f->after_mesh[i], f->after_tri[i], mesh, tri)))) return false;
	}
	if (f->callback && !f->callback(f->user, i, cand)) return false;
	return true;
}

/* rtk.c:550-566. kz is the FIRST axis whose |d| equals the maximum (x, then y,
 * then z); the sign mask uses sign BITS, so -0.0f counts as negative (rtk.c:152-154). */
void ora_ray_setup(const rtk_ray *ray, uint32_t k[3], float shear[3], uint32_t *sign_mask)
{
	float ax, ay, az, m;
	uint32_t bits[3], kz;
	memcpy(bits, ray->direction.v, sizeof(bits));
	{
		uint32_t a;
		a = bits[0] & 0x7fffffffu; memcpy(&ax, &a, 4);
		a = bits[1] & 0x7fffffffu; memcpy(&ay, &a, 4);
		a = bits[2] & 0x7fffffffu; memcpy(&az, &a, 4);
	}
	/* _rtk_maxcomp3, rtk.c:148-151: max(max(x,y) , z) with SSE operand order */
	m = sse_max(sse_max(ax, ay), az);
	kz = ax == m ? 0u : ay == m ? 1u : 2u;
	k[2] = kz;
	k[0] = (kz + 1u) % 3u;
	k[1] = (kz + 2u) % 3u;
	shear[0] = -ray->direction.v[k[0]] / ray->direction.v[kz];
	shear[1] = -ray->direction.v[k[1]] / ray->direction.v[kz];
	shear[2] = 1.0f / ray->direction.v[kz];
	*sign_mask = (bits[0] >> 31) | ((bits[1] >> 31) << 1) | ((bits[2] >> 31) << 2);
}

static inline bool id_less(uint32_t mesh_a, uint32_t tri_a, uint32_t mesh_b, uint32_t tri_b)
{
	return mesh_a < mesh_b || (mesh_a == mesh_b && tri_a < tri_b);
}

/* rtk.c:181-388: four triangle slots per step, padding slots included. */
static void leaf_visit(ora_trace *rt, uint64_t leaf_ptr)
{
	const char *const blob = rt->blob;
	const char *const leaf = blob + (leaf_ptr ^ 1u);
	uint64_t info;
	memcpy(&info, leaf, 8);
	const size_t num = (size_t)(info & 0x3fu);                 /* rtk.c:188 */
	const size_t num4 = (num + 3u) & ~(size_t)3u;                /* rtk.c:189 */
	const ora_leaf_tri *tris = (const ora_leaf_tri *)(leaf + 8);
	const uint32_t *mesh_table = (const uint32_t *)(tris + num4); /* rtk.c:192 */
	const rtk_vertex *verts = (const rtk_vertex *)(blob + (size_t)(info & ~(uint64_t)0x3f)); /* rtk.c:193 */

	const uint32_t kx = rt->k[0], ky = rt->k[1], kz = rt->k[2];
	const float sox = rt->so[0], soy = rt->so[1], soz = rt->so[2];
	const float shx = rt->sh[0], shy = rt->sh[1], shz = rt->sh[2];
	const float min_t = rt->ray.min_t;
	float max_t = rt->hit.t;                                     /* rtk.c:209 */

	if (rt->ctr) rt->ctr->leaves++;

	for (size_t g = 0; g < num4; g += 4) {
		float x[3][4], y[3][4], z[3][4];
		float u[4], v[4], w[4];
		bool any_zero = false;
		if (rt->ctr) rt->ctr->tri_groups++;

		for (int l = 0; l < 4; l++) {
			for (int c = 0; c < 3; c++) {
				const rtk_vertex *p = &verts[tris[g + l].v[c]];
				/* permute to shear space and move the origin, rtk.c:232-280 */
				const float vx = p->position.v[kx] - sox;
				const float vy = p->position.v[ky] - soy;
				const float vz = p->position.v[kz] - soz;
				/* shear, rtk.c:284-292: add(v, mul(s, vz)) */
				x[c][l] = vx + shx * vz;
				y[c][l] = vy + shy * vz;
				z[c][l] = shz * vz;
			}
			/* edge functions, rtk.c:298-300 */
			u[l] = x[1][l] * y[2][l] - y[1][l] * x[2][l];
			v[l] = x[2][l] * y[0][l] - y[2][l] * x[0][l];
			w[l] = x[0][l] * y[1][l] - y[0][l] * x[1][l];
			if (u[l] == 0.0f || v[l] == 0.0f || w[l] == 0.0f) any_zero = true;
		}

		/* rtk.c:302-336: if ANY lane has an exact zero, all four lanes are
		 * recomputed in double and rounded back to float. */
		if (any_zero) {
			for (int l = 0; l < 4; l++) {
				const double xd0 = x[0][l], yd0 = y[0][l];
				const double xd1 = x[1][l], yd1 = y[1][l];
				const double xd2 = x[2][l], yd2 = y[2][l];
				u[l] = (float)(xd1 * yd2 - yd1 * xd2);
				v[l] = (float)(xd2 * yd0 - yd2 * xd0);
				w[l] = (float)(xd0 * yd1 - yd0 * xd1);
			}
		}

		bool bad[4];
		int num_bad = 0;
		for (int l = 0; l < 4; l++) {
			/* rtk.c:340-342 */
			const bool neg = sse_min(sse_min(u[l], v[l]), w[l]) < 0.0f;
			const bool pos = sse_max(sse_max(u[l], v[l]), w[l]) > 0.0f;
			bad[l] = neg && pos;
			num_bad += bad[l];
		}
		if (num_bad == 4) continue;                               /* rtk.c:344 */

		for (int l = 0; l < 4; l++) {
			/* rtk.c:346-353 */
			const float det = (u[l] + v[l]) + w[l];
			const float rcp_det = 1.0f / det;
			float zz = u[l] * z[0][l];
			zz = zz + v[l] * z[1][l];
			zz = zz + w[l] * z[2][l];
			const float t = zz * rcp_det;
			if (bad[l]) continue;
			const ora_leaf_tri *tri = &tris[g + l];
			bool accept;
			if (rt->ties == ORA_TIES_REFERENCE) {
				/* rtk.c:354 then rtk.c:371 */
				accept = (t > min_t && t < max_t) && t < rt->hit.t;
			} else {
				accept = false;
				if (t > min_t && t < rt->ray.max_t) {
					if (t < rt->hit.t) accept = true;
					else if (rt->found && t == rt->hit.t) {
						accept = id_less(mesh_table[tri->local_mesh], tri->triangle_index,
							rt->hit.mesh_index, rt->hit.triangle_index);
					}
				}
			}
			if (accept && rt->filter) {
				rtk_hit cand;
				cand.t = t; cand.u = u[l] * rcp_det; cand.v = v[l] * rcp_det;
				cand.vertex[0] = verts[tri->v[0]]; cand.vertex[1] = verts[tri->v[1]]; cand.vertex[2] = verts[tri->v[2]];
				cand.mesh_index = mesh_table[tri->local_mesh];
				cand.triangle_index = tri->triangle_index;
				accept = filter_accepts(rt, cand.mesh_index, cand.triangle_index, t, &cand);
			}
			if (accept) {
				/* rtk.c:372-381 */
				rt->hit.t = t;
				rt->hit.u = u[l] * rcp_det;
				rt->hit.v = v[l] * rcp_det;
				rt->hit.vertex[0] = verts[tri->v[0]];
				rt->hit.vertex[1] = verts[tri->v[1]];
				rt->hit.vertex[2] = verts[tri->v[2]];
				rt->hit.mesh_index = mesh_table[tri->local_mesh];
				rt->hit.triangle_index = tri->triangle_index;
				rt->found = true;
				max_t = t;
			}
		}
	}
}

#define ORA_STACK_CAP (3 * ORA_MAX_DEPTH + 8)

/* rtk.c:390-539 with the push fix B5 (second/third/fourth nearest children land at
 * depth-1/-2/-3 so that the nearest one is popped first). The two sentinel stack
 * slots of the reference are replaced by an explicit empty check. */
static void bvh_traverse(ora_trace *rt, uint64_t root_ptr)
{
	const char *const blob = rt->blob;
	float stack_t[ORA_STACK_CAP];
	uint64_t stack_ptr[ORA_STACK_CAP];
	uint32_t depth = 0;
	float top_t = -RTK_INF;
	uint64_t top_ptr = root_ptr;

	const float ox = rt->ray.origin.x, oy = rt->ray.origin.y, oz = rt->ray.origin.z;
	const float rdx = 1.0f / rt->ray.direction.x;                /* rtk.c:410, RTK_MM_RCP is a true divide */
	const float rdy = 1.0f / rt->ray.direction.y;
	const float rdz = 1.0f / rt->ray.direction.z;
	const float ray_min_t = rt->ray.min_t;
	const uint32_t sx = rt->sign_mask & 1u, sy = (rt->sign_mask >> 1) & 1u, sz = rt->sign_mask >> 2;

	for (;;) {
		const float hit_t = rt->hit.t;

		/* rtk.c:432-437 */
		while (rt->ties == ORA_TIES_REFERENCE ? top_t >= hit_t : top_t > hit_t) {
			if (depth == 0) return;
			--depth;
			top_t = stack_t[depth];
			top_ptr = stack_ptr[depth];
		}

		if (top_ptr & 1u) {                                       /* rtk.c:441-447 */
			leaf_visit(rt, top_ptr);
			if (depth == 0) return;
			--depth;
			top_t = stack_t[depth];
			top_ptr = stack_ptr[depth];
			continue;
		}

		const ora_node *node = (const ora_node *)(blob + (size_t)top_ptr);
		float ts[4];
		uint32_t mask_bits = 0, num_bits = 0;
		if (rt->ctr) rt->ctr->nodes++;
		for (int i = 0; i < 4; i++) {
			/* rtk.c:458-465: (bound - origin) * rcp, near/far picked by sign bit */
			const float min_x = (node->bx[sx][i] - ox) * rdx;
			const float max_x = (node->bx[sx ^ 1u][i] - ox) * rdx;
			const float min_y = (node->by[sy][i] - oy) * rdy;
			const float max_y = (node->by[sy ^ 1u][i] - oy) * rdy;
			const float min_z = (node->bz[sz][i] - oz) * rdz;
			const float max_z = (node->bz[sz ^ 1u][i] - oz) * rdz;
			const float tmin = sse_max(sse_max(min_x, min_y), sse_max(min_z, ray_min_t));
			const float tmax = sse_min(sse_min(max_x, max_y), sse_min(max_z, hit_t));
			if (tmin <= tmax) {                                   /* rtk.c:470 */
				ts[i] = tmin;
				mask_bits |= 1u << i;
				num_bits++;
			} else {
				ts[i] = RTK_INF;                                  /* rtk.c:471 */
			}
		}

		if (mask_bits == 0) {                                     /* rtk.c:475-480 */
			if (depth == 0) return;
			--depth;
			top_t = stack_t[depth];
			top_ptr = stack_ptr[depth];
		} else if (num_bits == 1) {                               /* rtk.c:481-488 */
			uint32_t bit = 0;
			while (!((mask_bits >> bit) & 1u)) bit++;
			top_t = sse_min(sse_min(ts[0], ts[1]), sse_min(ts[2], ts[3]));
			top_ptr = node->child[bit];
		} else {
			/* rtk.c:496-517: slot number in the two low mantissa bits, sort ascending.
			 * The four tagged values are distinct, so any correct sort equals the network. */
			uint32_t tag[4];
			for (int i = 0; i < 4; i++) {
				uint32_t b;
				memcpy(&b, &ts[i], 4);
				tag[i] = (b & ~3u) | (uint32_t)i;
			}
			for (int i = 1; i < 4; i++) {
				uint32_t kb = tag[i];
				float kf; memcpy(&kf, &kb, 4);
				int j = i - 1;
				for (; j >= 0; j--) {
					float jf; memcpy(&jf, &tag[j], 4);
					if (!(kf < jf)) break;
					tag[j + 1] = tag[j];
				}
				tag[j + 1] = kb;
			}
			float st[4];
			uint32_t si[4];
			for (int i = 0; i < 4; i++) {
				uint32_t b = tag[i] & ~3u;
				memcpy(&st[i], &b, 4);
				si[i] = tag[i] & 3u;
			}
			/* rtk.c:520-535 with B5: nearest becomes top, the others are pushed far to near */
			for (uint32_t i = num_bits - 1; i >= 1; i--) {
				stack_t[depth] = st[i];
				stack_ptr[depth] = node->child[si[i]];
				depth++;
			}
			top_t = st[0];
			top_ptr = node->child[si[0]];
		}
	}
}

bool ora_trace_ray(const void *blob, const rtk_ray *ray, rtk_hit *hit, int ties, ora_counters *ctr)
{
	ora_trace rt;
	memset(&rt, 0, sizeof(rt));
	rt.blob = (const char *)blob;
	rt.ray = *ray;
	rt.hit.t = ray->max_t;                                        /* rtk.c:548 */
	rt.found = false;
	rt.ties = ties;
	rt.ctr = ctr;
	ora_ray_setup(ray, rt.k, rt.sh, &rt.sign_mask);
	rt.so[0] = ray->origin.v[rt.k[0]];
	rt.so[1] = ray->origin.v[rt.k[1]];
	rt.so[2] = ray->origin.v[rt.k[2]];
	if (ctr) ctr->rays++;

	bvh_traverse(&rt, ORA_ROOT_OFFSET);                           /* rtk.c:569 */

	if (rt.hit.t < ray->max_t) {                                  /* rtk.c:571-576 */
		*hit = rt.hit;
		if (ctr) ctr->hits++;
		return true;
	}
	return false;
}

void ora_trace_rays(const void *blob, const rtk_ray *rays, size_t n, rtk_hit *hits,
	uint8_t *mask, int ties, int threads, ora_counters *total)
{
	ora_counters sum;
	memset(&sum, 0, sizeof(sum));
	if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
	{
		ora_counters local;
		memset(&local, 0, sizeof(local));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1024)
#endif
		for (long long i = 0; i < (long long)n; i++) {
			rtk_hit h;
			bool ok = ora_trace_ray(blob, &rays[i], &h, ties, total ? &local : NULL);
			if (ok && hits) hits[i] = h;
			if (mask) mask[i] = ok ? 1 : 0;
		}
#ifdef _OPENMP
#pragma omp critical
#endif
		{
			sum.rays += local.rays; sum.nodes += local.nodes; sum.leaves += local.leaves;
			sum.tri_groups += local.tri_groups; sum.hits += local.hits;
		}
	}
	if (total) *total = sum;
}

/* ---- the CPU baseline's timing driver (bench.py, cpu_baseline) --------------------------------------------------
 * The same per-ray code as ora_trace_rays; what differs is everything AROUND it that decided the earlier figures
 * (46.6 Mrays/s on 16 threads, 33.4 on 128): the 68-byte rtk_hit output was allocated and first touched inside the timed
 * call -- 1.1 GB of page faults under one address-space lock -- and handed out in dynamic chunks of 64. Here the output is
 * a 16-byte record per ray (what the GPU writes) in a buffer the caller allocated and touched beforehand, rays are
 * dealt in chunks of 1024 (16 k grabs for 2^24 rays: no contention, and a slow core -- an SMT sibling, a busy neighbour -- does not hold the others up), and the blob can be copied into memory whose pages were first touched by the worker
 * threads round robin (ora_alloc_spread), so that on a multi-socket host it does not sit on one memory controller. */
void ora_trace_rays_records(const void *blob, const rtk_ray *rays, size_t n, ora_record *out, int ties, int threads)
{
	if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
#endif
	for (long long i = 0; i < (long long)n; i++) {
		rtk_hit h;
		ora_record r;
		if (ora_trace_ray(blob, &rays[i], &h, ties, NULL)) { r.t = h.t; r.u = h.u; r.v = h.v; r.triangle_index = h.triangle_index; }
		else { r.t = rays[i].max_t; r.u = 0.0f; r.v = 0.0f; r.triangle_index = 0xffffffffu; }
		out[i] = r;
	}
}

/* `size` bytes, 4096-aligned, every page first touched (zeroed) by thread (page % threads) of an OpenMP team; with
 * src != NULL the pages are filled from there instead. Free with ora_free. */
void *ora_alloc_spread(size_t size, const void *src, int threads)
{
	void *p = NULL;
	const size_t page = 4096, bytes = align_up(size ? size : 1, page);
	if (threads < 1) threads = 1;
	if (posix_memalign(&p, page, bytes) != 0) return NULL;
	const long long pages = (long long)(bytes / page);
#ifdef _OPENMP
#pragma omp parallel for schedule(static, 1) num_threads(threads)
#endif
	for (long long k = 0; k < pages; k++) {
		const size_t at = (size_t)k * page;
		const size_t len = at + page <= size ? page : (at < size ? size - at : 0);
		if (src && len) memcpy((char *)p + at, (const char *)src + at, len);
		if (len < page) memset((char *)p + at + len, 0, page - len);
	}
	return p;
}

/* Canonical ties; the filter is asked about a candidate exactly when it would become the new closest hit, so
 * the result is the closest candidate the filter accepts (the order of the questions depends on the BVH, the
 * answer does not, as long as the filter is a function of the candidate). */
void ora_trace_rays_filter(const void *blob, const rtk_ray *rays, size_t n, rtk_hit *hits, uint8_t *mask,
	int threads, const ora_filter *filter)
{
	if (threads < 1) threads = 1;
	if (filter && filter->callback) threads = 1;   /* callbacks may come from an interpreter */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
#endif
	for (long long i = 0; i < (long long)n; i++) {
		ora_trace rt;
		const rtk_ray *ray = &rays[i];
		memset(&rt, 0, sizeof(rt));
		rt.blob = (const char *)blob;
		rt.ray = *ray;
		rt.hit.t = ray->max_t;
		rt.ties = ORA_TIES_CANONICAL;
		rt.filter = filter;
		rt.ray_index = (size_t)i;
		ora_ray_setup(ray, rt.k, rt.sh, &rt.sign_mask);
		rt.so[0] = ray->origin.v[rt.k[0]];
		rt.so[1] = ray->origin.v[rt.k[1]];
		rt.so[2] = ray->origin.v[rt.k[2]];
		bvh_traverse(&rt, ORA_ROOT_OFFSET);
		/* with a filter "found" (not t < max_t) says whether a candidate was accepted */
		if (rt.found && hits) hits[i] = rt.hit;
		if (mask) mask[i] = rt.found ? 1 : 0;
	}
}

void ora_trace_chain(const void *const *blobs, size_t num_blobs, const rtk_ray *rays, size_t n,
	rtk_hit *hits, uint8_t *mask, int threads)
{
	if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
#endif
	for (long long i = 0; i < (long long)n; i++) {
		rtk_ray r = rays[i];
		rtk_hit best;
		bool any = false;
		for (size_t b = 0; b < num_blobs; b++) {
			rtk_hit h;
			if (ora_trace_ray(blobs[b], &r, &h, ORA_TIES_REFERENCE, NULL)) {
				best = h;
				any = true;
				r.max_t = h.t;
			}
		}
		if (any && hits) hits[i] = best;
		if (mask) mask[i] = any ? 1 : 0;
	}
}

/* ------------------------------------------------------------------------- */
/* Blob emission helpers                                                      */
/* ------------------------------------------------------------------------- */

static void write_header(rtk_scene *s, uint64_t size, uint64_t node_off, uint64_t leaf_off, uint64_t vert_off)
{
	/* rtk.c:1737-1755 */
	static const char magic[8] = { 0, 'R', 'T', 'K', '\r', '\n', 0x1a, '\n' };
	memcpy(s->magic, magic, 8);
	s->endian = 0xaabb;
	s->sizeof_real = (uint8_t)sizeof(rtk_real);
	s->pad_0 = 0;
	s->version = 1;
	s->pad_1 = 0;
	s->size_in_bytes = size;
	s->node_offset = node_off;
	s->leaf_offset = leaf_off;
	s->vertex_offset = vert_off;
}

static void set_empty_slot(ora_node *n, int i, uint64_t null_leaf_offset)
{
	/* rtk.c:1612-1620; the leaf tag is added (B19) */
	n->bx[0][i] = +1.0f; n->bx[1][i] = -1.0f;
	n->by[0][i] = +1.0f; n->by[1][i] = -1.0f;
	n->bz[0][i] = +1.0f; n->bz[1][i] = -1.0f;
	n->child[i] = null_leaf_offset | 1u;
}

size_t ora_make_leaf_blob(const rtk_vertex *verts, const uint32_t *mesh_index,
	const uint32_t *triangle_index, size_t n, void *dst, size_t cap)
{
	if (n > 63) return 0;
	const size_t n4 = (n + 3u) & ~(size_t)3u;
	uint32_t meshes[63];
	size_t num_meshes = 0;
	uint8_t local[63];
	for (size_t i = 0; i < n; i++) {
		size_t m = 0;
		for (; m < num_meshes; m++) if (meshes[m] == mesh_index[i]) break;
		if (m == num_meshes) meshes[num_meshes++] = mesh_index[i];
		local[i] = (uint8_t)m;
	}
	const size_t node_off = 128;
	const size_t leaf_off = node_off + sizeof(ora_node);            /* 256 */
	const size_t null_leaf = leaf_off;                              /* 64 bytes, info = 0 */
	const size_t the_leaf = leaf_off + 64;
	const size_t leaf_bytes = align_up(8 + 8 * n4 + 4 * num_meshes, 64);
	const size_t vert_off = align_up(the_leaf + leaf_bytes, 128);
	const size_t total = align_up(vert_off + 16 * (n ? 3 * n : 1), 128);
	if (cap < total) return 0;

	char *b = (char *)dst;
	memset(b, 0, total);
	write_header((rtk_scene *)b, total, node_off, leaf_off, vert_off);

	ora_node *root = (ora_node *)(b + node_off);
	float mn[3] = { RTK_INF, RTK_INF, RTK_INF }, mx[3] = { -RTK_INF, -RTK_INF, -RTK_INF };
	for (size_t i = 0; i < 3 * n; i++)
		for (int a = 0; a < 3; a++) {
			mn[a] = sse_min(mn[a], verts[i].position.v[a]);
			mx[a] = sse_max(mx[a], verts[i].position.v[a]);
		}
	for (int i = 0; i < 4; i++) set_empty_slot(root, i, null_leaf);
	if (n) {
		root->bx[0][0] = mn[0]; root->bx[1][0] = mx[0];
		root->by[0][0] = mn[1]; root->by[1][0] = mx[1];
		root->bz[0][0] = mn[2]; root->bz[1][0] = mx[2];
		root->child[0] = the_leaf | 1u;
	}
	uint64_t info = (uint64_t)n | (uint64_t)vert_off;
	memcpy(b + the_leaf, &info, 8);
	ora_leaf_tri *tris = (ora_leaf_tri *)(b + the_leaf + 8);
	for (size_t i = 0; i < n; i++) {
		tris[i].v[0] = (uint8_t)(3 * i);
		tris[i].v[1] = (uint8_t)(3 * i + 1);
		tris[i].v[2] = (uint8_t)(3 * i + 2);
		tris[i].local_mesh = local[i];
		tris[i].triangle_index = triangle_index[i];
	}
	memcpy(tris + n4, meshes, 4 * num_meshes);
	memcpy(b + vert_off, verts, 16 * 3 * n);
	return total;
}

/* ------------------------------------------------------------------------- */
/* Builder                                                                    */
/* ------------------------------------------------------------------------- */

#define NONE ((size_t)-1)

typedef struct {                  /* rtk.c:598-604 */
	float mn[3], mx[3];
	rtk_vertex v[3];
	uint32_t mesh_index;
	uint32_t triangle_index;
	uint8_t v_ix[3];
} b_item;

typedef struct {                  /* rtk.c:606-613 */
	float mn[3], mx[3];
	size_t begin, count;
	size_t child;                 /* index of the first of two children, or NONE */
	size_t vgroup;                /* first vertex of the node's vertex group, or NONE */
	uint32_t depth;
	/* linearisation scratch */
	uint64_t out_offset;
} b_node;

typedef struct {
	rtk_scene_desc desc;
	size_t num_tris;
	b_item *items;
	b_node *nodes;
	size_t nodes_cap, nodes_used;
	size_t vertex_count;
	size_t leaf_bytes;
	float item_cost, split_cost;  /* never initialised in v0 (B9): 1 and 1 */
} builder;

static void log_line(builder *b, const char *s)
{
	if (b->desc.log_fn) b->desc.log_fn(b->desc.log_user, NULL, s);
}

/* rtk.c:1028-1070 */
static void decode_indices(const rtk_mesh *m, uint32_t *dst, size_t offset, size_t count)
{
	if (m->index_cb) { m->index_cb(m->index_cb_user, m, dst, offset, count); return; }
	if (m->index.data) {
		if (m->index.type == RTK_TYPE_U16) {
			const size_t stride = m->index.stride ? m->index.stride : 6;
			for (size_t i = 0; i < count; i++) {
				const uint16_t *ix = (const uint16_t *)((const char *)m->index.data + (offset + i) * stride);
				dst[3 * i] = ix[0]; dst[3 * i + 1] = ix[1]; dst[3 * i + 2] = ix[2];
			}
		} else { /* U32 and DEFAULT (rtk.h:69 says the default is U32) */
			const size_t stride = m->index.stride ? m->index.stride : 12;
			for (size_t i = 0; i < count; i++) {
				const uint32_t *ix = (const uint32_t *)((const char *)m->index.data + (offset + i) * stride);
				dst[3 * i] = ix[0]; dst[3 * i + 1] = ix[1]; dst[3 * i + 2] = ix[2];
			}
		}
	} else {
		for (size_t i = 0; i < count; i++) {
			const uint32_t base = (uint32_t)(offset + i) * 3u;
			dst[3 * i] = base; dst[3 * i + 1] = base + 1; dst[3 * i + 2] = base + 2;
		}
	}
}

/* rtk.c:1072-1114; the F64 branch reads doubles (B20) */
static void decode_vertices(const rtk_mesh *m, rtk_vec3 *dst, const uint32_t *idx, size_t count)
{
	if (m->position_cb) { m->position_cb(m->position_cb_user, m, dst, idx, count); return; }
	const char *data = (const char *)m->position.data;
	if (m->position.type == RTK_TYPE_F64) {
		const size_t stride = m->position.stride ? m->position.stride : 24;
		for (size_t i = 0; i < 3 * count; i++) {
			const double *p = (const double *)(data + (size_t)idx[i] * stride);
			dst[i].x = (float)p[0]; dst[i].y = (float)p[1]; dst[i].z = (float)p[2];
		}
	} else {
		const size_t stride = m->position.stride ? m->position.stride : 12;
		for (size_t i = 0; i < 3 * count; i++) {
			const float *p = (const float *)(data + (size_t)idx[i] * stride);
			dst[i].x = p[0]; dst[i].y = p[1]; dst[i].z = p[2];
		}
	}
}

/* rtk.c:1116-1182: items in concatenated mesh order, triangle_index per mesh */
static void setup_triangles(builder *b, float mn[3], float mx[3])
{
	b_item *item = b->items;
	for (int a = 0; a < 3; a++) { mn[a] = RTK_INF; mx[a] = -RTK_INF; }
	for (size_t mi = 0; mi < b->desc.num_meshes; mi++) {
		const rtk_mesh *m = &b->desc.meshes[mi];
		size_t offset = 0, left = m->num_triangles;
		while (left > 0) {
			const size_t chunk = left > 128 ? 128 : left;
			uint32_t indices[128 * 3];
			rtk_vec3 vertices[128 * 3 + 1];
			decode_indices(m, indices, offset, chunk);
			decode_vertices(m, vertices, indices, chunk);
			for (size_t i = 0; i < chunk; i++) {
				for (int a = 0; a < 3; a++) {
					float lo = vertices[3 * i].v[a], hi = lo;
					lo = sse_min(lo, vertices[3 * i + 1].v[a]); hi = sse_max(hi, vertices[3 * i + 1].v[a]);
					lo = sse_min(lo, vertices[3 * i + 2].v[a]); hi = sse_max(hi, vertices[3 * i + 2].v[a]);
					item->mn[a] = lo; item->mx[a] = hi;
					mn[a] = sse_min(mn[a], lo); mx[a] = sse_max(mx[a], hi);
				}
				for (int c = 0; c < 3; c++) {
					item->v[c].position = vertices[3 * i + c];
					item->v[c].index = indices[3 * i + c];
				}
				item->mesh_index = (uint32_t)mi;
				item->triangle_index = (uint32_t)(offset + i);
				item++;
			}
			offset += chunk;
			left -= chunk;
		}
	}
}

static float bounds_area(const float mn[3], const float mx[3])
{
	/* rtk.c:729-733 */
	const float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
	return 2.0f * (x * y + y * z + z * x);
}

static int largest_axis(const b_node *n)
{
	/* intent of rtk.c:773-775 (B10): largest extent of max - min */
	const float sx = n->mx[0] - n->mn[0], sy = n->mx[1] - n->mn[1], sz = n->mx[2] - n->mn[2];
	const float m = sse_max(sse_max(sx, sy), sz);
	return sx == m ? 0 : sy == m ? 1 : 2;
}

static int g_sort_axis; /* builder is single threaded */
static int item_cmp(const void *pa, const void *pb)
{
	/* rtk.c:739-755 with a real three-way result and an id tiebreak (B22) */
	const b_item *a = (const b_item *)pa, *b = (const b_item *)pb;
	const float ma = a->mn[g_sort_axis] + a->mx[g_sort_axis];
	const float mb = b->mn[g_sort_axis] + b->mx[g_sort_axis];
	if (ma < mb) return -1;
	if (ma > mb) return +1;
	if (a->mesh_index != b->mesh_index) return a->mesh_index < b->mesh_index ? -1 : +1;
	if (a->triangle_index != b->triangle_index) return a->triangle_index < b->triangle_index ? -1 : +1;
	return 0;
}

static void make_leaf(builder *b, b_node *n)
{
	/* rtk.c:765-811 */
	n->child = NONE;
	n->vgroup = NONE;
	g_sort_axis = largest_axis(n);
	qsort(b->items + n->begin, n->count, sizeof(b_item), item_cmp);
	uint32_t uniq[ORA_LEAF_MAX_ITEMS + 1];
	size_t nu = 0;
	for (size_t i = 0; i < n->count; i++) {
		const uint32_t m = b->items[n->begin + i].mesh_index;
		size_t k = 0;
		for (; k < nu; k++) if (uniq[k] == m) break;
		if (k == nu) uniq[nu++] = m;
	}
	const size_t n4 = (n->count + 3u) & ~(size_t)3u;
	b->leaf_bytes += align_up(8 + 8 * n4 + 4 * nu, 64);
}

static size_t alloc_children(builder *b)
{
	const size_t c = b->nodes_used;
	b->nodes_used += 2;
	return c;
}

static void child_init(b_node *c, size_t begin, size_t count, uint32_t depth)
{
	c->begin = begin; c->count = count; c->child = NONE; c->vgroup = NONE; c->depth = depth; c->out_offset = 0;
	for (int a = 0; a < 3; a++) { c->mn[a] = RTK_INF; c->mx[a] = -RTK_INF; }
}

static void grow(float mn[3], float mx[3], const float imn[3], const float imx[3])
{
	for (int a = 0; a < 3; a++) { mn[a] = sse_min(mn[a], imn[a]); mx[a] = sse_max(mx[a], imx[a]); }
}

static void build_node(builder *b, size_t node_index);

static void split_equal(builder *b, size_t node_index)
{
	/* rtk.c:813-865 */
	b_node *n = &b->nodes[node_index];
	g_sort_axis = largest_axis(n);
	b_item *items = b->items + n->begin;
	qsort(items, n->count, sizeof(b_item), item_cmp);
	const size_t nl = n->count / 2, nr = n->count - nl;
	const size_t ci = alloc_children(b);
	n = &b->nodes[node_index];
	n->child = ci;
	b_node *c = &b->nodes[ci];
	child_init(&c[0], n->begin, nl, n->depth + 1);
	child_init(&c[1], n->begin + nl, nr, n->depth + 1);
	for (size_t i = 0; i < nl; i++) grow(c[0].mn, c[0].mx, items[i].mn, items[i].mx);
	for (size_t i = nl; i < n->count; i++) grow(c[1].mn, c[1].mx, items[i].mn, items[i].mx);
	build_node(b, ci);
	build_node(b, ci + 1);
}

typedef struct { float mn[3], mx[3], rmn[3], rmx[3]; uint32_t num; } b_bin;

static int bin_of(float mid2, float min2, float scale)
{
	/* rtk.c:899-902 */
	int k = (int)((mid2 - min2) * scale);
	if (k < 0) k = 0;
	if (k > ORA_BINS - 1) k = ORA_BINS - 1;
	return k;
}

static void split_sah(builder *b, size_t node_index)
{
	/* rtk.c:867-1019 */
	b_node *n = &b->nodes[node_index];
	b_item *items = b->items + n->begin;
	b_bin bins[ORA_BINS];
	float best_cost = RTK_INF;
	int best_axis = -1, best_bin = 0;
	float best_mn[2][3], best_mx[2][3];
	const float rcp_parent_area = 1.0f / bounds_area(n->mn, n->mx);

	for (int axis = 0; axis < 3; axis++) {
		const float lo = n->mn[axis], hi = n->mx[axis];
		if (!(hi > lo)) continue; /* guards the 0*inf of rtk.c:893 on flat nodes */
		for (int i = 0; i < ORA_BINS; i++) {
			for (int a = 0; a < 3; a++) { bins[i].mn[a] = RTK_INF; bins[i].mx[a] = -RTK_INF; }
			bins[i].num = 0;
		}
		const float min2 = lo + lo;
		const float scale = (0.5f * (float)ORA_BINS) / (hi - lo);
		for (size_t i = 0; i < n->count; i++) {
			b_bin *bn = &bins[bin_of(items[i].mn[axis] + items[i].mx[axis], min2, scale)];
			grow(bn->mn, bn->mx, items[i].mn, items[i].mx);
			bn->num++;
		}
		/* suffix bounds, rtk.c:909-915 */
		memcpy(bins[ORA_BINS - 1].rmn, bins[ORA_BINS - 1].mn, 12);
		memcpy(bins[ORA_BINS - 1].rmx, bins[ORA_BINS - 1].mx, 12);
		for (int i = ORA_BINS - 1; i > 0; i--)
			for (int a = 0; a < 3; a++) {
				bins[i - 1].rmn[a] = sse_min(bins[i - 1].mn[a], bins[i].rmn[a]);
				bins[i - 1].rmx[a] = sse_max(bins[i - 1].mx[a], bins[i].rmx[a]);
			}
		/* prefix scan, rtk.c:917-945 */
		float lmn[3] = { RTK_INF, RTK_INF, RTK_INF }, lmx[3] = { -RTK_INF, -RTK_INF, -RTK_INF };
		size_t nl = 0;
		for (int i = 0; i < ORA_BINS - 1; i++) {
			grow(lmn, lmx, bins[i].mn, bins[i].mx);
			nl += bins[i].num;
			const size_t nr = n->count - nl;
			if (nl == 0 || nr == 0) continue;
			const float al = bounds_area(lmn, lmx);
			const float ar = bounds_area(bins[i + 1].rmn, bins[i + 1].rmx);
			const float cl = (float)((nl + 3) / 4) * b->item_cost;
			const float cr = (float)((nr + 3) / 4) * b->item_cost;
			const float cost = b->split_cost + (al * cl + ar * cr) * rcp_parent_area;
			if (cost < best_cost) {
				memcpy(best_mn[0], lmn, 12); memcpy(best_mx[0], lmx, 12);
				memcpy(best_mn[1], bins[i + 1].rmn, 12); memcpy(best_mx[1], bins[i + 1].rmx, 12);
				best_cost = cost; best_axis = axis; best_bin = i;
			}
		}
	}

	const float leaf_cost = (float)n->count * b->item_cost;
	if (best_cost < leaf_cost || n->count > ORA_LEAF_MAX_ITEMS) {
		if (best_axis < 0) {
			/* intent of rtk.c:952-958 (B11): too big -> equal split, else leaf */
			if (n->count > ORA_LEAF_MAX_ITEMS) split_equal(b, node_index);
			else make_leaf(b, n);
			return;
		}
		/* partition with the same bin function, rtk.c:963-986 */
		const float lo = n->mn[best_axis], hi = n->mx[best_axis];
		const float min2 = lo + lo;
		const float scale = (0.5f * (float)ORA_BINS) / (hi - lo);
		b_item *first = items, *last = items + n->count;
		while (first != last) {
			if (bin_of(first->mn[best_axis] + first->mx[best_axis], min2, scale) <= best_bin) {
				first++;
			} else {
				last--;
				b_item tmp = *first; *first = *last; *last = tmp;
			}
		}
		const size_t nl = (size_t)(first - items), nr = n->count - nl;
		const size_t ci = alloc_children(b);
		n = &b->nodes[node_index];
		n->child = ci;
		b_node *c = &b->nodes[ci];
		child_init(&c[0], n->begin, nl, n->depth + 1);
		child_init(&c[1], n->begin + nl, nr, n->depth + 1);
		memcpy(c[0].mn, best_mn[0], 12); memcpy(c[0].mx, best_mx[0], 12);
		memcpy(c[1].mn, best_mn[1], 12); memcpy(c[1].mx, best_mx[1], 12);
		build_node(b, ci);
		build_node(b, ci + 1);
	} else {
		make_leaf(b, n);
	}
}

static void build_node(builder *b, size_t node_index)
{
	/* rtk.c:1421-1453 */
	b_node *n = &b->nodes[node_index];
	if (n->depth >= ORA_MAX_DEPTH) { make_leaf(b, n); return; }
	uint64_t splits_left = ORA_MAX_DEPTH - n->depth - 1;
	if (splits_left > 63) splits_left = 63;
	if (((uint64_t)n->count >> splits_left) > ORA_LEAF_MAX_ITEMS) { split_equal(b, node_index); return; }
	if (n->count <= ORA_LEAF_MIN_ITEMS) { make_leaf(b, n); return; }
	split_sah(b, node_index);
}

/* -- vertex groups, rtk.c:1186-1360 -- */

typedef struct { uint64_t e[ORA_VSET_MAX]; size_t size; } vset;

static bool vset_insert(vset *s, uint32_t mesh, uint32_t vix)
{
	/* rtk.c:1196-1214 (sorted insert, no duplicates) */
	const uint64_t key = ((uint64_t)mesh << 32) | vix;
	size_t lo = 0, hi = s->size;
	while (lo < hi) { size_t mid = (lo + hi) / 2; if (s->e[mid] < key) lo = mid + 1; else hi = mid; }
	if (lo < s->size && s->e[lo] == key) return true;
	if (s->size == ORA_VSET_MAX) return false;
	memmove(&s->e[lo + 1], &s->e[lo], (s->size - lo) * 8);
	s->e[lo] = key;
	s->size++;
	return true;
}

static bool vset_merge(vset *d, const vset *a, const vset *b)
{
	/* rtk.c:1216-1245 */
	size_t i = 0, j = 0, k = 0;
	while (i < a->size || j < b->size) {
		uint64_t v;
		if (j == b->size || (i < a->size && a->e[i] <= b->e[j])) {
			v = a->e[i];
			if (j < b->size && b->e[j] == v) j++;
			i++;
		} else {
			v = b->e[j++];
		}
		if (k == ORA_VSET_MAX) return false;
		d->e[k++] = v;
	}
	d->size = k;
	return true;
}

static uint32_t vset_find(const vset *s, uint32_t mesh, uint32_t vix)
{
	/* rtk.c:1254-1280 */
	const uint64_t key = ((uint64_t)mesh << 32) | vix;
	size_t lo = 0, hi = s->size;
	while (lo < hi) { size_t mid = (lo + hi) / 2; if (s->e[mid] < key) lo = mid + 1; else hi = mid; }
	return (uint32_t)lo;
}

static void assign_vertices(builder *b, b_node *n, const vset *s, size_t off)
{
	/* rtk.c:1282-1301 with the SIZE_MAX comparison of B14 */
	if (n->vgroup != NONE) return;
	n->vgroup = off;
	if (n->child != NONE) {
		assign_vertices(b, &b->nodes[n->child], s, off);
		assign_vertices(b, &b->nodes[n->child + 1], s, off);
	} else {
		for (size_t i = 0; i < n->count; i++) {
			b_item *it = &b->items[n->begin + i];
			for (int c = 0; c < 3; c++) it->v_ix[c] = (uint8_t)vset_find(s, it->mesh_index, it->v[c].index);
		}
	}
}

static void close_vertices(builder *b, b_node *n, const vset *s)
{
	/* rtk.c:1305-1309; groups start on a 64-byte boundary = 4 vertices (B16) */
	b->vertex_count = align_up(b->vertex_count, 4);
	const size_t off = b->vertex_count;
	b->vertex_count += s->size;
	assign_vertices(b, n, s, off);
}

static bool gather_vertices(builder *b, b_node *n, vset *out)
{
	/* rtk.c:1313-1360 */
	if (n->child != NONE) {
		b_node *c = &b->nodes[n->child];
		vset s[2];
		s[0].size = s[1].size = 0;
		const bool o0 = gather_vertices(b, &c[0], &s[0]);
		const bool o1 = gather_vertices(b, &c[1], &s[1]);
		if (!o0 && !o1) return false;
		if (o0 && o1) {
			if (vset_merge(out, &s[0], &s[1])) return true;
			const int close_ix = s[1].size > s[0].size ? 1 : 0;
			close_vertices(b, &c[close_ix], &s[close_ix]);
			*out = s[close_ix ^ 1];
			return true;
		}
		*out = s[o1 ? 1 : 0];
		return true;
	}
	for (size_t i = 0; i < n->count; i++) {
		const b_item *it = &b->items[n->begin + i];
		for (int c = 0; c < 3; c++) vset_insert(out, it->mesh_index, it->v[c].index);
	}
	return true;
}

static void finalize_node(builder *b, size_t node_index)
{
	/* rtk.c:1484-1507 */
	b_node *n = &b->nodes[node_index];
	if (n->count >= 2048 && n->child != NONE) {
		finalize_node(b, n->child);
		finalize_node(b, n->child + 1);
	} else {
		vset root;
		root.size = 0;
		if (gather_vertices(b, n, &root)) close_vertices(b, n, &root);
	}
}

/* -- linearisation, rtk.c:1509-1622 -- */

/* children of the 4-wide node made from binary node src (rtk.c:1572-1592) */
static void wide_children(builder *b, const b_node *src, b_node *out[4])
{
	for (unsigned i = 0; i < 4; i++) {
		b_node *mid = &b->nodes[src->child + (i >> 1)];
		b_node *c;
		if (mid->child != NONE) c = &b->nodes[mid->child + (i & 1u)];
		else c = (i & 1u) == 0 ? mid : NULL;
		if (c && c->count == 0) c = NULL;
		out[i] = c;
	}
}

static void write_leaf(builder *b, char *blob, uint64_t leaf_at, uint64_t vertex_section, const b_node *src, uint64_t *leaf_size_out)
{
	/* intent of rtk.c:1509-1568 (B16, B17, B18, B24) */
	const size_t n = src->count, n4 = (n + 3u) & ~(size_t)3u;
	const uint64_t group_bytes = vertex_section + (uint64_t)src->vgroup * 16u;
	const uint64_t info = (uint64_t)n | group_bytes;
	memcpy(blob + leaf_at, &info, 8);
	ora_leaf_tri *tris = (ora_leaf_tri *)(blob + leaf_at + 8);
	uint32_t *mesh_table = (uint32_t *)(tris + n4);
	size_t nu = 0;
	rtk_vertex *verts = (rtk_vertex *)(blob + group_bytes);
	for (size_t i = 0; i < n; i++) {
		const b_item *it = &b->items[src->begin + i];
		for (int c = 0; c < 3; c++) {
			tris[i].v[c] = it->v_ix[c];
			verts[it->v_ix[c]] = it->v[c];
		}
		tris[i].triangle_index = it->triangle_index;
		size_t k = 0;
		for (; k < nu; k++) if (mesh_table[k] == it->mesh_index) break;
		if (k == nu) mesh_table[nu++] = it->mesh_index;
		tris[i].local_mesh = (uint8_t)k;
	}
	*leaf_size_out = align_up(8 + 8 * n4 + 4 * nu, 64);
}

void *ora_build_scene(const rtk_scene_desc *desc, size_t *size_out)
{
	builder bs, *b = &bs;
	memset(b, 0, sizeof(*b));
	b->desc = *desc;
	b->item_cost = 1.0f;
	b->split_cost = 1.0f;
	for (size_t i = 0; i < desc->num_meshes; i++) b->num_tris += desc->meshes[i].num_triangles;

	/* rtk.c:1643-1658: items + binary nodes (2N-1 bound, +3 for a virtual root) */
	b->items = (b_item *)malloc(sizeof(b_item) * (b->num_tris ? b->num_tris : 1));
	b->nodes_cap = 2 * b->num_tris + 4;
	b->nodes = (b_node *)calloc(b->nodes_cap, sizeof(b_node));
	if (!b->items || !b->nodes) { free(b->items); free(b->nodes); return NULL; }

	log_line(b, "oracle: gathering triangles");
	b_node *root = &b->nodes[0];
	child_init(root, 0, b->num_tris, 0);
	setup_triangles(b, root->mn, root->mx);                       /* rtk.c:1362-1419 */
	b->nodes_used = 1;
	b->leaf_bytes = 64;                                           /* null leaf, rtk.c:1677 */

	log_line(b, "oracle: building nodes");
	build_node(b, 0);

	/* rtk.c:1460-1476: a root that is a leaf gets a virtual parent (B15: two distinct children) */
	root = &b->nodes[0];
	if (root->child == NONE) {
		const size_t ci = alloc_children(b);
		b_node *c = &b->nodes[ci];
		c[0] = *root;
		c[0].depth = 1;
		child_init(&c[1], 0, 0, 1);
		memcpy(c[1].mn, root->mn, 12); memcpy(c[1].mx, root->mn, 12);
		root->child = ci;
		root->vgroup = NONE;
	}

	log_line(b, "oracle: finalizing nodes");
	finalize_node(b, 0);

	/* count and place the 4-wide nodes breadth first (intent of B13, B23) */
	size_t wide_cap = b->nodes_used / 2 + 2, wide_num = 0, head = 0;
	b_node **queue = (b_node **)malloc(sizeof(b_node *) * wide_cap);
	if (!queue) { free(b->items); free(b->nodes); return NULL; }
	queue[wide_num++] = &b->nodes[0];
	while (head < wide_num) {
		b_node *kids[4];
		wide_children(b, queue[head++], kids);
		for (int i = 0; i < 4; i++) if (kids[i] && kids[i]->child != NONE) queue[wide_num++] = kids[i];
	}

	/* rtk.c:1719-1730, 1745-1755 */
	const uint64_t node_off = 128;
	const uint64_t leaf_off = align_up(node_off + wide_num * sizeof(ora_node), 128);
	const uint64_t vert_off = align_up(leaf_off + b->leaf_bytes, 128);
	const uint64_t total = align_up(vert_off + align_up(b->vertex_count, 4) * 16u, 128);
	char *blob = (char *)alloc_aligned(total);
	if (!blob) { free(queue); free(b->items); free(b->nodes); return NULL; }
	write_header((rtk_scene *)blob, total, node_off, leaf_off, vert_off);

	for (size_t i = 0; i < wide_num; i++) queue[i]->out_offset = node_off + i * sizeof(ora_node);
	uint64_t leaf_cursor = leaf_off + 64;                         /* after the null leaf, rtk.c:1763-1765 */
	for (size_t qi = 0; qi < wide_num; qi++) {
		/* rtk.c:1570-1622 */
		ora_node *dst = (ora_node *)(blob + queue[qi]->out_offset);
		b_node *kids[4];
		wide_children(b, queue[qi], kids);
		for (int i = 0; i < 4; i++) {
			const b_node *c = kids[i];
			if (!c) { set_empty_slot(dst, i, leaf_off); continue; }
			dst->bx[0][i] = c->mn[0]; dst->bx[1][i] = c->mx[0];
			dst->by[0][i] = c->mn[1]; dst->by[1][i] = c->mx[1];
			dst->bz[0][i] = c->mn[2]; dst->bz[1][i] = c->mx[2];
			if (c->child != NONE) {
				dst->child[i] = c->out_offset;
			} else {
				uint64_t sz;
				dst->child[i] = leaf_cursor | 1u;
				write_leaf(b, blob, leaf_cursor, vert_off, c, &sz);
				leaf_cursor += sz;
			}
		}
	}

	free(queue);
	free(b->items);
	free(b->nodes);
	if (size_out) *size_out = (size_t)total;
	return blob;
}

/* ------------------------------------------------------------------------- */
/* Validation                                                                 */
/* ------------------------------------------------------------------------- */

int ora_validate_blob(const void *blobv, size_t size, uint64_t *num_nodes, uint64_t *num_leaves, uint64_t *num_tris)
{
	const char *blob = (const char *)blobv;
	const rtk_scene *s = (const rtk_scene *)blob;
	static const char magic[8] = { 0, 'R', 'T', 'K', '\r', '\n', 0x1a, '\n' };
	uint64_t nn = 0, nl = 0, nt = 0;
	if (size < 256) return -1;
	if (memcmp(s->magic, magic, 8) != 0) return -2;
	if (s->endian != 0xaabb || s->sizeof_real != 4 || s->version != 1) return -3;
	if (s->size_in_bytes > size || s->node_offset != 128) return -4;
	/* iterative walk */
	size_t cap = 1024, top = 0;
	uint64_t *stack = (uint64_t *)malloc(cap * 8);
	if (!stack) return -9;
	stack[top++] = ORA_ROOT_OFFSET;
	int rc = 0;
	while (top && rc == 0) {
		const uint64_t p = stack[--top];
		if (p & 1u) {
			const uint64_t at = p ^ 1u;
			if (at + 8 > s->size_in_bytes) { rc = -5; break; }
			uint64_t info; memcpy(&info, blob + at, 8);
			const uint64_t n = info & 0x3f, n4 = (n + 3) & ~3ull, vg = info & ~0x3full;
			if (at + 8 + 8 * n4 > s->size_in_bytes) { rc = -5; break; }
			if (n && vg + 16 > s->size_in_bytes) { rc = -6; break; }
			const ora_leaf_tri *tris = (const ora_leaf_tri *)(blob + at + 8);
			for (uint64_t i = 0; i < n4; i++)
				for (int c = 0; c < 3; c++)
					if (vg + 16ull * (tris[i].v[c] + 1u) > s->size_in_bytes) rc = -6;
			for (uint64_t i = n; i < n4; i++)
				if (tris[i].v[0] | tris[i].v[1] | tris[i].v[2]) rc = -7; /* padding slots must be zero */
			if (n) { nl++; nt += n; }
		} else {
			if (p + sizeof(ora_node) > s->size_in_bytes || (p & 3u)) { rc = -8; break; }
			const ora_node *n = (const ora_node *)(blob + p);
			nn++;
			for (int i = 0; i < 4; i++) {
				if (!(n->bx[0][i] <= n->bx[1][i])) continue; /* empty slot */
				if (top == cap) {
					cap *= 2;
					uint64_t *ns = (uint64_t *)realloc(stack, cap * 8);
					if (!ns) { rc = -9; break; }
					stack = ns;
				}
				stack[top++] = n->child[i];
			}
		}
	}
	free(stack);
	if (num_nodes) *num_nodes = nn;
	if (num_leaves) *num_leaves = nl;
	if (num_tris) *num_tris = nt;
	return rc;
}
