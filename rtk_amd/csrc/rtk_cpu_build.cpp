// rtk_cpu_build.cpp -- the reference's caller-scheduled build as a working CPU task graph
// (reference rtk.h:108-120, rtk.c:1362-1507, 867-1019, 1186-1360, 1509-1622).
//
// The reference hands the application a graph of tasks to run on its own thread pool: rtk_start_build
// returns the first task, rtk_run_task runs one task and appends the tasks it spawned to a caller-owned
// queue, and the build is complete when nothing is pending. At v0 that graph cannot finish (SURVEY.md
// appendix B); this is the same graph with the defects' INTENT implemented:
//   phase A  split the triangles into <= 128 ranges; one task per range decodes indices and positions
//            (buffers or callbacks, <= 128 triangles per callback, rtk.c:1116-1182) into build items
//   phase B  one task per node: binned SAH (32 bins x 3 axes, cost model of rtk.c:931-949 with working
//            constants, B9), in-place partition, two child tasks; small subtrees are finished inside
//            their task
//   phase C  vertex groups: (mesh, vertex) pairs de-duplicated bottom-up into groups of <= 256 so that
//            leaves can use 8-bit vertex indices (rtk.c:1186-1360)
//   finish   two binary levels -> one 4-wide node (rtk.c:1570-1622), blob emission (appendix A)
// Phases are separated by a counter of pending tasks; whichever thread retires the last task of a phase
// starts the next one (rtk.c:1701-1714). rtk_run_task may be called from any number of threads.
//
// This is host code only and an ALTERNATIVE builder a host selects explicitly (rtk_amd_set_builder or
// RTK_AMD_BUILDER=cpu): it produces a scene blob; tracing that blob is still the GPU's job
// (rtk_trace_ray[s] upload it), and nothing falls back to it on its own.
#include "rtk.h"
#include "rtk_amd.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

void rtk_set_error(const char *fmt, ...);

namespace {

const size_t MAX_RANGES = 128;            // RTK_MAX_CONCURRENT_TASKS, rtk.c:590-592
const size_t MIN_RANGE = 1024;            // rtk.c:1370
const int SAH_BINS = 32;                  // RTK_BUILD_SPLITS, rtk.c:586-588
const size_t LEAF_MAX = 63;               // 6-bit count in the leaf header (rtk.c:188; B7)
const uint32_t MAX_DEPTH = 64;            // RTK_BVH_MAX_DEPTH, rtk.c:5
const size_t GROUP_MAX = 256;             // RTK_VERTEX_SET_MAX_SIZE, rtk.c:1186
const size_t INLINE_ITEMS = 4096;         // subtrees up to this size are built inside their task
const size_t FINALIZE_SPLIT = 2048;       // rtk.c:1492: finalize recurses into tasks above this size
// cost constants of rtk.c:931-949 (1 / 1 as the oracle uses). RTK_AMD_CPU_SAH_SPLIT_COST /
// RTK_AMD_CPU_SAH_ITEM_COST change them for experiments (a split cost of 0.5 gives the one-triangle leaves of the device
// build: bench.py --bvh cpu-sah); read once per process.
float sah_env(const char *name, float dflt)
{
	const char *v = getenv(name);
	if (!v || !*v) return dflt;
	const float f = (float)atof(v);
	return f > 0.0f ? f : dflt;
}
const size_t LEAF_MIN = (size_t)sah_env("RTK_AMD_CPU_LEAF_MIN", 4.0f);       // RTK_BVH_LEAF_MIN_ITEMS, rtk.c:6 (1: split down to single triangles)
const float SAH_ITEM_COST = sah_env("RTK_AMD_CPU_SAH_ITEM_COST", 1.0f), SAH_SPLIT_COST = sah_env("RTK_AMD_CPU_SAH_SPLIT_COST", 1.0f);

struct Item {
	float mn[3], mx[3];
	float pos[3][3];
	uint32_t vidx[3];
	uint32_t mesh, tri;
	uint8_t local[3];                     // index into the vertex group, set in phase C
};

struct Node {
	float mn[3], mx[3];
	size_t begin, count;
	int64_t left, right;                  // node indices, -1 = leaf
	uint32_t depth;
	int64_t group;                        // vertex group of this LEAF
};

struct Group {
	std::vector<uint64_t> keys;           // sorted (mesh << 32 | vertex index)
	std::vector<rtk_vertex> verts;        // same order
	uint64_t byte_offset;                 // in the vertex section
};

struct RangeState { size_t base, num; float mn[3], mx[3]; };

} // namespace

struct rtk_task_ctx {
	rtk_task *queue;
	size_t capacity, num;
};

struct rtk_cpu_build {
	rtk_scene_desc desc;
	std::vector<uint64_t> mesh_base;
	size_t num_triangles = 0;
	std::vector<Item> items;
	std::vector<Node> nodes;
	std::atomic<size_t> nodes_used{ 0 };
	std::atomic<size_t> tasks_left{ 0 };
	std::atomic<int> phase{ 0 };          // 0 setup, 1 nodes, 2 finalize, 3 done
	std::vector<RangeState> ranges;
	std::mutex group_mutex;
	std::vector<Group *> groups;
	void *owner = nullptr;                // the rtk_build that wraps this
	// finish
	bool planned = false;
	std::vector<int64_t> wide_root;       // binary node of each 4-wide node, breadth first
	std::vector<int64_t> wide_child;      // 4 per wide node: >= 0 binary node of an inner child, -2 empty, -1-leafidx... see plan()
	std::vector<int64_t> leaf_nodes;      // binary leaf nodes in emission order
	std::vector<uint64_t> leaf_offset;
	uint64_t node_off = 128, leaf_off = 0, vert_off = 0, total = 0;
};

namespace {

void log_line(rtk_cpu_build *b, const char *fmt, size_t a, size_t c)
{
	if (!b->desc.log_fn) return;
	char buf[256];
	snprintf(buf, sizeof(buf), fmt, a, c);
	b->desc.log_fn(b->desc.log_user, (rtk_build *)b->owner, buf);
}

// ---- task plumbing (rtk.c:700-710): a task that does not fit the caller's queue is run in place

void task_setup_range(const rtk_task *t, rtk_task_ctx *ctx);
void task_build_node(const rtk_task *t, rtk_task_ctx *ctx);
void task_finalize_node(const rtk_task *t, rtk_task_ctx *ctx);

void push_task(rtk_cpu_build *b, rtk_task_ctx *ctx, rtk_task_fn *fn, size_t index, double cost)
{
	rtk_task t;
	t.build = (rtk_build *)b->owner;
	t.fn = fn;
	t.cost = cost;
	t.index = index;
	t.arg = (uintptr_t)b;
	if (ctx->num < ctx->capacity) {
		b->tasks_left.fetch_add(1);
		ctx->queue[ctx->num++] = t;
	} else {
		fn(&t, ctx);                       // queue full (the reference would write past it, B21)
	}
}

// ---- phase A: triangle setup (rtk.c:1028-1182)

void decode_chunk(const rtk_mesh *m, size_t offset, size_t count, uint32_t *indices, rtk_vec3 *verts)
{
	if (m->index_cb) m->index_cb(m->index_cb_user, m, indices, offset, count);
	else if (m->index.data) {
		const bool u16 = m->index.type == RTK_TYPE_U16;
		const size_t stride = m->index.stride ? m->index.stride : (u16 ? 6 : 12);
		for (size_t i = 0; i < count; i++) {
			const char *p = (const char *)m->index.data + (offset + i) * stride;
			for (int c = 0; c < 3; c++) indices[3 * i + c] = u16 ? ((const uint16_t *)p)[c] : ((const uint32_t *)p)[c];
		}
	} else {
		for (size_t i = 0; i < count; i++) for (int c = 0; c < 3; c++) indices[3 * i + c] = (uint32_t)((offset + i) * 3 + c);   // rtk.c:1061-1068
	}
	if (m->position_cb) m->position_cb(m->position_cb_user, m, verts, indices, count);
	else {
		const bool f64 = m->position.type == RTK_TYPE_F64;
		const size_t stride = m->position.stride ? m->position.stride : (f64 ? 24 : 12);
		for (size_t i = 0; i < 3 * count; i++) {
			const char *p = (const char *)m->position.data + (size_t)indices[i] * stride;
			if (f64) { verts[i].x = (float)((const double *)p)[0]; verts[i].y = (float)((const double *)p)[1]; verts[i].z = (float)((const double *)p)[2]; }   // B20
			else { verts[i].x = ((const float *)p)[0]; verts[i].y = ((const float *)p)[1]; verts[i].z = ((const float *)p)[2]; }
		}
	}
}

void task_setup_range(const rtk_task *t, rtk_task_ctx *)
{
	rtk_cpu_build *b = (rtk_cpu_build *)t->arg;
	RangeState &rs = b->ranges[t->index];
	for (int a = 0; a < 3; a++) { rs.mn[a] = INFINITY; rs.mx[a] = -INFINITY; }
	log_line(b, "rtk_amd cpu build: triangle range %zu (%zu triangles)", t->index, rs.num);
	// the range is a run of global triangle numbers; walk the meshes it touches in chunks of <= 128
	size_t g = rs.base, left = rs.num;
	size_t mi = (size_t)(std::upper_bound(b->mesh_base.begin(), b->mesh_base.end(), (uint64_t)g) - b->mesh_base.begin()) - 1;
	while (left) {
		while (b->mesh_base[mi + 1] <= g) mi++;
		const rtk_mesh *m = &b->desc.meshes[mi];
		const size_t in_mesh = g - (size_t)b->mesh_base[mi];
		size_t chunk = std::min<size_t>(std::min<size_t>(left, 128), (size_t)b->mesh_base[mi + 1] - g);
		uint32_t indices[128 * 3];
		rtk_vec3 verts[128 * 3 + 1];
		decode_chunk(m, in_mesh, chunk, indices, verts);
		for (size_t i = 0; i < chunk; i++) {
			Item &it = b->items[g + i];
			for (int a = 0; a < 3; a++) { it.mn[a] = INFINITY; it.mx[a] = -INFINITY; }
			for (int c = 0; c < 3; c++) {
				const float p[3] = { verts[3 * i + c].x, verts[3 * i + c].y, verts[3 * i + c].z };
				for (int a = 0; a < 3; a++) { it.pos[c][a] = p[a]; it.mn[a] = std::min(it.mn[a], p[a]); it.mx[a] = std::max(it.mx[a], p[a]); }
				it.vidx[c] = indices[3 * i + c];
			}
			it.mesh = (uint32_t)mi;
			it.tri = (uint32_t)(in_mesh + i);             // per-mesh triangle index, rtk.c:1168-1169
			for (int a = 0; a < 3; a++) { rs.mn[a] = std::min(rs.mn[a], it.mn[a]); rs.mx[a] = std::max(rs.mx[a], it.mx[a]); }
		}
		g += chunk;
		left -= chunk;
	}
}

// ---- phase B: binned SAH (rtk.c:867-1019)

float half_area(const float mn[3], const float mx[3])
{
	const float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
	return x * y + y * z + z * x;
}

void make_leaf(Node &n) { n.left = n.right = -1; }

size_t alloc_pair(rtk_cpu_build *b)
{
	const size_t at = b->nodes_used.fetch_add(2);       // children are adjacent (rtk.c:828, 992)
	return at + 1 < b->nodes.size() ? at : (size_t)-1;
}

void fit_bounds(const rtk_cpu_build *b, Node &n)
{
	for (int a = 0; a < 3; a++) { n.mn[a] = INFINITY; n.mx[a] = -INFINITY; }
	for (size_t i = n.begin; i < n.begin + n.count; i++)
		for (int a = 0; a < 3; a++) { n.mn[a] = std::min(n.mn[a], b->items[i].mn[a]); n.mx[a] = std::max(n.mx[a], b->items[i].mx[a]); }
}

// Split node `ni` (or make it a leaf). Returns true and the two children if it was split.
bool split_node(rtk_cpu_build *b, size_t ni, size_t *left, size_t *right)
{
	Node &n = b->nodes[ni];
	if (n.count <= LEAF_MIN || n.depth + 1 >= MAX_DEPTH) {
		if (n.count <= LEAF_MAX) { make_leaf(n); return false; }
	}
	Item *items = b->items.data();
	// centroid bounds (x2, like the reference: min + max, rtk.c:890-907)
	float cmn[3] = { INFINITY, INFINITY, INFINITY }, cmx[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (size_t i = n.begin; i < n.begin + n.count; i++)
		for (int a = 0; a < 3; a++) { const float c = items[i].mn[a] + items[i].mx[a]; cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c); }
	const float parent_area = half_area(n.mn, n.mx);
	float best_cost = INFINITY;
	int best_axis = -1, best_bin = -1;
	for (int a = 0; a < 3; a++) {
		const float ext = cmx[a] - cmn[a];
		if (!(ext > 0.0f)) continue;
		const float scale = (float)SAH_BINS / ext;
		float bmn[SAH_BINS][3], bmx[SAH_BINS][3];
		uint32_t cnt[SAH_BINS];
		for (int k = 0; k < SAH_BINS; k++) { cnt[k] = 0; for (int q = 0; q < 3; q++) { bmn[k][q] = INFINITY; bmx[k][q] = -INFINITY; } }
		for (size_t i = n.begin; i < n.begin + n.count; i++) {
			int k = (int)(((items[i].mn[a] + items[i].mx[a]) - cmn[a]) * scale);
			k = k < 0 ? 0 : (k >= SAH_BINS ? SAH_BINS - 1 : k);
			cnt[k]++;
			for (int q = 0; q < 3; q++) { bmn[k][q] = std::min(bmn[k][q], items[i].mn[q]); bmx[k][q] = std::max(bmx[k][q], items[i].mx[q]); }
		}
		// suffix scan (rtk.c:909-915), then prefix scan with the cost of rtk.c:931-936
		float rarea[SAH_BINS];
		uint32_t rcnt[SAH_BINS];
		float amn[3] = { INFINITY, INFINITY, INFINITY }, amx[3] = { -INFINITY, -INFINITY, -INFINITY };
		uint32_t acc = 0;
		for (int k = SAH_BINS - 1; k >= 1; k--) {
			if (cnt[k]) for (int q = 0; q < 3; q++) { amn[q] = std::min(amn[q], bmn[k][q]); amx[q] = std::max(amx[q], bmx[k][q]); }
			acc += cnt[k];
			rcnt[k] = acc;
			rarea[k] = acc ? half_area(amn, amx) : 0.0f;
		}
		for (int q = 0; q < 3; q++) { amn[q] = INFINITY; amx[q] = -INFINITY; }
		acc = 0;
		for (int k = 0; k < SAH_BINS - 1; k++) {
			if (cnt[k]) for (int q = 0; q < 3; q++) { amn[q] = std::min(amn[q], bmn[k][q]); amx[q] = std::max(amx[q], bmx[k][q]); }
			acc += cnt[k];
			if (acc == 0 || rcnt[k + 1] == 0) continue;
			const float nl = (float)((acc + 3) / 4), nr = (float)((rcnt[k + 1] + 3) / 4);        // groups of four (rtk.c:933-934)
			const float cost = SAH_SPLIT_COST + (half_area(amn, amx) * nl + rarea[k + 1] * nr) * SAH_ITEM_COST / (parent_area > 0.0f ? parent_area : 1.0f);
			if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
		}
	}
	const float leaf_cost = (float)n.count * SAH_ITEM_COST;
	if (n.count <= LEAF_MAX && !(best_cost < leaf_cost)) { make_leaf(n); return false; }       // rtk.c:948-949
	size_t mid;
	if (best_axis >= 0) {
		const float scale = (float)SAH_BINS / (cmx[best_axis] - cmn[best_axis]);
		Item *lo = items + n.begin, *hi = items + n.begin + n.count;
		Item *m = std::partition(lo, hi, [&](const Item &it) {
			int k = (int)(((it.mn[best_axis] + it.mx[best_axis]) - cmn[best_axis]) * scale);
			k = k < 0 ? 0 : (k >= SAH_BINS ? SAH_BINS - 1 : k);
			return k <= best_bin;
		});
		mid = (size_t)(m - items);
	} else mid = n.begin;
	if (mid == n.begin || mid == n.begin + n.count) {
		// no usable plane (all centroids equal): equal split along the largest extent (rtk.c:813-865, B10/B11)
		int axis = 0;
		for (int a = 1; a < 3; a++) if (n.mx[a] - n.mn[a] > n.mx[axis] - n.mn[axis]) axis = a;
		mid = n.begin + n.count / 2;
		std::nth_element(items + n.begin, items + mid, items + n.begin + n.count,
			[&](const Item &x, const Item &y) { return x.mn[axis] + x.mx[axis] < y.mn[axis] + y.mx[axis]; });
	}
	const size_t pair = alloc_pair(b);
	if (pair == (size_t)-1) { make_leaf(n); return false; }                                     // cannot happen: 2N nodes are reserved
	Node &l = b->nodes[pair], &r = b->nodes[pair + 1];
	l.begin = n.begin; l.count = mid - n.begin; l.depth = n.depth + 1; l.group = -1;
	r.begin = mid; r.count = n.begin + n.count - mid; r.depth = n.depth + 1; r.group = -1;
	fit_bounds(b, l);
	fit_bounds(b, r);
	n.left = (int64_t)pair;
	n.right = (int64_t)pair + 1;
	*left = pair;
	*right = pair + 1;
	return true;
}

void build_subtree(rtk_cpu_build *b, size_t ni)
{
	size_t l, r;
	if (!split_node(b, ni, &l, &r)) return;
	build_subtree(b, l);
	build_subtree(b, r);
}

void task_build_node(const rtk_task *t, rtk_task_ctx *ctx)
{
	rtk_cpu_build *b = (rtk_cpu_build *)t->arg;
	const size_t ni = t->index;
	if (b->nodes[ni].count <= INLINE_ITEMS) { build_subtree(b, ni); return; }
	size_t l, r;
	if (!split_node(b, ni, &l, &r)) return;
	push_task(b, ctx, task_build_node, l, 10.0 * (double)b->nodes[l].count);    // rtk.c:861-864
	push_task(b, ctx, task_build_node, r, 10.0 * (double)b->nodes[r].count);
}

// ---- phase C: vertex groups (rtk.c:1186-1360)

uint64_t vkey(const Item &it, int c) { return ((uint64_t)it.mesh << 32) | it.vidx[c]; }

// closes the subtree under `ni` as ONE group holding exactly `keys`
void close_group(rtk_cpu_build *b, size_t ni, std::vector<uint64_t> &keys)
{
	Group *g = new Group();
	g->keys.swap(keys);
	g->verts.resize(g->keys.size());
	g->byte_offset = 0;
	const Node &n = b->nodes[ni];
	for (size_t i = n.begin; i < n.begin + n.count; i++) {
		Item &it = b->items[i];
		for (int c = 0; c < 3; c++) {
			const size_t k = (size_t)(std::lower_bound(g->keys.begin(), g->keys.end(), vkey(it, c)) - g->keys.begin());
			it.local[c] = (uint8_t)k;
			rtk_vertex &v = g->verts[k];
			v.position.x = it.pos[c][0]; v.position.y = it.pos[c][1]; v.position.z = it.pos[c][2];
			v.index = it.vidx[c];
		}
	}
	int64_t id;
	{
		std::lock_guard<std::mutex> lock(b->group_mutex);
		id = (int64_t)b->groups.size();
		b->groups.push_back(g);
	}
	// every leaf below refers to this group
	std::vector<size_t> stack(1, ni);
	while (!stack.empty()) {
		const size_t k = stack.back();
		stack.pop_back();
		Node &m = b->nodes[k];
		if (m.left < 0) m.group = id;
		else { stack.push_back((size_t)m.left); stack.push_back((size_t)m.right); }
	}
}

// Returns the sorted distinct vertex keys of the subtree if it is still OPEN (may merge into its parent's
// group); an empty vector with *closed = true once the subtree has been given groups of its own.
std::vector<uint64_t> gather_group(rtk_cpu_build *b, size_t ni, bool *closed)
{
	const Node &n = b->nodes[ni];
	std::vector<uint64_t> keys;
	*closed = false;
	if (n.left < 0) {
		keys.reserve(3 * n.count);
		for (size_t i = n.begin; i < n.begin + n.count; i++) for (int c = 0; c < 3; c++) keys.push_back(vkey(b->items[i], c));
		std::sort(keys.begin(), keys.end());
		keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
		return keys;                                                       // <= 189 keys: always fits a group
	}
	bool lc, rc;
	std::vector<uint64_t> lk = gather_group(b, (size_t)n.left, &lc), rk = gather_group(b, (size_t)n.right, &rc);
	if (!lc && !rc) {
		keys.resize(lk.size() + rk.size());
		keys.erase(std::set_union(lk.begin(), lk.end(), rk.begin(), rk.end(), keys.begin()), keys.end());
		if (keys.size() <= GROUP_MAX) return keys;                         // still open
	}
	if (!lc) close_group(b, (size_t)n.left, lk);
	if (!rc) close_group(b, (size_t)n.right, rk);
	*closed = true;
	return std::vector<uint64_t>();
}

void task_finalize_node(const rtk_task *t, rtk_task_ctx *ctx)
{
	rtk_cpu_build *b = (rtk_cpu_build *)t->arg;
	const size_t ni = t->index;
	const Node &n = b->nodes[ni];
	if (n.left >= 0 && n.count >= FINALIZE_SPLIT) {
		// large subtrees cannot share one group anyway (> 256 distinct vertices in practice): their halves are
		// independent tasks (rtk.c:1492-1500)
		push_task(b, ctx, task_finalize_node, (size_t)n.left, 5.0 * (double)b->nodes[(size_t)n.left].count);
		push_task(b, ctx, task_finalize_node, (size_t)n.right, 5.0 * (double)b->nodes[(size_t)n.right].count);
		return;
	}
	bool closed;
	std::vector<uint64_t> keys = gather_group(b, ni, &closed);
	if (!closed) close_group(b, ni, keys);
}

// ---- phase starters (rtk.c:1362-1391, 1393-1419, 1455-1482)

void start_phase(rtk_cpu_build *b, int phase, rtk_task_ctx *ctx)
{
	if (phase == 0) {
		const size_t n = b->num_triangles;
		size_t count = (n + MIN_RANGE - 1) / MIN_RANGE;
		count = count < 1 ? 1 : (count > MAX_RANGES ? MAX_RANGES : count);
		b->ranges.resize(count);
		for (size_t r = 0; r < count; r++) {
			const size_t lo = n * r / count, hi = n * (r + 1) / count;
			b->ranges[r].base = lo;
			b->ranges[r].num = hi - lo;
		}
		log_line(b, "rtk_amd cpu build: %zu triangles in %zu setup tasks", n, count);
		for (size_t r = 0; r < count; r++) push_task(b, ctx, task_setup_range, r, 10.0 * (double)b->ranges[r].num);
	} else if (phase == 1) {
		Node &root = b->nodes[0];
		root.begin = 0; root.count = b->num_triangles; root.depth = 0; root.left = root.right = -1; root.group = -1;
		for (int a = 0; a < 3; a++) { root.mn[a] = INFINITY; root.mx[a] = -INFINITY; }
		for (const RangeState &rs : b->ranges) if (rs.num) for (int a = 0; a < 3; a++) { root.mn[a] = std::min(root.mn[a], rs.mn[a]); root.mx[a] = std::max(root.mx[a], rs.mx[a]); }
		b->nodes_used.store(1);
		log_line(b, "rtk_amd cpu build: node phase, root of %zu items (%zu ranges merged)", root.count, b->ranges.size());
		push_task(b, ctx, task_build_node, 0, 10.0 * (double)root.count);
	} else if (phase == 2) {
		log_line(b, "rtk_amd cpu build: finalize phase, %zu binary nodes (%zu triangles)", b->nodes_used.load(), b->num_triangles);
		push_task(b, ctx, task_finalize_node, 0, 5.0 * (double)b->num_triangles);
	}
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) & ~(a - 1); }

// ---- finish: 4-wide collapse and blob layout (rtk.c:1570-1622, 1719-1774; appendix A)

// children of the 4-wide node rooted at binary node `ni`: child i = grandchild [i >> 1][i & 1]; a binary child that
// is a leaf goes into slot 2*side with slot 2*side+1 empty (rtk.c:1572-1592)
void wide_children(const rtk_cpu_build *b, int64_t ni, int64_t out[4])
{
	const Node &n = b->nodes[(size_t)ni];
	out[0] = out[1] = out[2] = out[3] = -2;                                 // -2 = empty slot
	if (n.left < 0) { out[0] = ni; return; }                                 // a root that is a single leaf
	const int64_t side[2] = { n.left, n.right };
	for (int s = 0; s < 2; s++) {
		const Node &c = b->nodes[(size_t)side[s]];
		if (c.left < 0) out[2 * s] = side[s];
		else { out[2 * s] = c.left; out[2 * s + 1] = c.right; }
	}
}

void plan(rtk_cpu_build *b)
{
	if (b->planned) return;
	b->wide_root.assign(1, 0);
	b->wide_child.clear();
	b->leaf_nodes.clear();
	for (size_t w = 0; w < b->wide_root.size(); w++) {
		int64_t c[4];
		wide_children(b, b->wide_root[w], c);
		for (int k = 0; k < 4; k++) {
			int64_t ref = -2;
			if (c[k] >= 0) {
				if (b->nodes[(size_t)c[k]].left < 0) { ref = -3 - (int64_t)b->leaf_nodes.size(); b->leaf_nodes.push_back(c[k]); }   // leaf L -> -3 - L
				else { ref = (int64_t)b->wide_root.size(); b->wide_root.push_back(c[k]); }
			}
			b->wide_child.push_back(ref);
		}
	}
	// vertex groups, 64-byte aligned (rtk.c:193)
	uint64_t vbytes = 0;
	for (Group *g : b->groups) { g->byte_offset = vbytes; vbytes += align_up(g->verts.size() * sizeof(rtk_vertex), 64); }
	// leaves: 64 bytes of null leaf first (rtk.c:1763-1765)
	uint64_t lbytes = 64;
	b->leaf_offset.resize(b->leaf_nodes.size());
	for (size_t l = 0; l < b->leaf_nodes.size(); l++) {
		const Node &n = b->nodes[(size_t)b->leaf_nodes[l]];
		std::vector<uint32_t> meshes;
		for (size_t i = n.begin; i < n.begin + n.count; i++) if (std::find(meshes.begin(), meshes.end(), b->items[i].mesh) == meshes.end()) meshes.push_back(b->items[i].mesh);
		b->leaf_offset[l] = lbytes;
		lbytes += align_up(8 + 8 * ((n.count + 3) & ~(size_t)3) + 4 * meshes.size(), 64);
	}
	b->leaf_off = align_up(b->node_off + b->wide_root.size() * 128, 128);
	b->vert_off = align_up(b->leaf_off + lbytes, 128);
	b->total = align_up(b->vert_off + vbytes, 128);
	b->planned = true;
}

} // namespace

// ---------------------------------------------------------------------------- entry points (called from rtk_build.hip)

// The first task of a build (rtk.c:1679-1681): it does nothing itself, running it opens phase A.
void rtk_cpu_task_start(const rtk_task *, rtk_task_ctx *) {}

rtk_cpu_build *rtk_cpu_build_start(const rtk_scene_desc *desc, void *owner, rtk_task *first_task, rtk_task_fn *runner)
{
	rtk_cpu_build *b = new rtk_cpu_build();
	b->desc = *desc;                                                       // meshes stay borrowed (rtk.c:1661)
	b->owner = owner;
	b->mesh_base.assign(desc->num_meshes + 1, 0);
	for (size_t m = 0; m < desc->num_meshes; m++) {
		const rtk_mesh *me = &desc->meshes[m];
		if (me->num_triangles && !me->position.data && !me->position_cb) { rtk_set_error("rtk_start_build: mesh %zu has no positions", m); delete b; return nullptr; }
		if (me->index.data && me->index.type != RTK_TYPE_U16 && me->index.type != RTK_TYPE_U32 && me->index.type != RTK_TYPE_DEFAULT) { rtk_set_error("rtk_start_build: bad index type"); delete b; return nullptr; }
		if (me->position.data && !me->position_cb && me->position.type != RTK_TYPE_DEFAULT && me->position.type != RTK_TYPE_REAL && me->position.type != RTK_TYPE_F32 &&
			me->position.type != RTK_TYPE_F64) { rtk_set_error("rtk_start_build: bad position type"); delete b; return nullptr; }      // rtk.c:1080-1113 asserts
		b->mesh_base[m + 1] = b->mesh_base[m] + me->num_triangles;
	}
	b->num_triangles = (size_t)b->mesh_base.back();
	if (b->num_triangles >= 0xfffffff0ull) { rtk_set_error("rtk_start_build: too many triangles"); delete b; return nullptr; }
	b->items.resize(b->num_triangles);
	b->nodes.resize(2 * std::max<size_t>(b->num_triangles, 2) + 2);
	b->tasks_left.store(1);
	b->phase.store(-1);
	(void)runner;
	if (first_task) {
		first_task->fn = rtk_cpu_task_start;
		first_task->build = (rtk_build *)owner;
		first_task->cost = 0.0;
		first_task->index = 0;
		first_task->arg = (uintptr_t)b;
	}
	return b;
}

// One task. `start` = this is the first task of the build (it only opens phase A).
size_t rtk_cpu_build_run(rtk_cpu_build *b, const rtk_task *task, bool start, rtk_task *queue, size_t queue_size)
{
	rtk_task_ctx ctx = { queue, queue_size, 0 };
	if (!start) task->fn(task, &ctx);
	// the thread that retires the last pending task of a phase starts the next one (rtk.c:1701-1714); a phase that
	// queues nothing (everything ran in place) falls through to the one after it
	while (b->tasks_left.fetch_sub(1) == 1) {
		const int next = b->phase.load() + 1;
		if (next > 2) { b->phase.store(3); break; }
		b->phase.store(next);
		b->tasks_left.fetch_add(1);                 // the starter itself counts as pending until it has queued its tasks
		start_phase(b, next, &ctx);
	}
	return ctx.num;
}

bool rtk_cpu_build_done(const rtk_cpu_build *b) { return b->phase.load() == 3; }

size_t rtk_cpu_build_size(rtk_cpu_build *b)
{
	if (!rtk_cpu_build_done(b)) { rtk_set_error("rtk_get_build_size: the build's tasks have not all run"); return 0; }
	plan(b);
	return (size_t)b->total;
}

bool rtk_cpu_build_write(rtk_cpu_build *b, void *buffer, size_t size)
{
	if (!rtk_cpu_build_done(b)) { rtk_set_error("rtk_finish_build: the build's tasks have not all run"); return false; }
	plan(b);
	if (size < b->total) return false;
	char *blob = (char *)buffer;
	memset(blob, 0, b->total);                                            // padding slots must read as zero (B24)
	rtk_scene *s = (rtk_scene *)blob;
	static const char magic[8] = { 0, 'R', 'T', 'K', '\r', '\n', 0x1a, '\n' };
	memcpy(s->magic, magic, 8);
	s->endian = 0xaabb; s->sizeof_real = 4; s->pad_0 = 0; s->version = 1; s->pad_1 = 0;
	s->size_in_bytes = b->total; s->node_offset = b->node_off; s->leaf_offset = b->leaf_off; s->vertex_offset = b->vert_off;
	for (size_t w = 0; w < b->wide_root.size(); w++) {
		float *bx = (float *)(blob + b->node_off + w * 128);             // bounds_x[2][4], bounds_y[2][4], bounds_z[2][4]
		uint64_t *ptr = (uint64_t *)(blob + b->node_off + w * 128 + 96);
		int64_t c[4];
		wide_children(b, b->wide_root[w], c);
		for (int k = 0; k < 4; k++) {
			const int64_t ref = b->wide_child[4 * w + k];
			if (ref == -2) {
				for (int a = 0; a < 3; a++) { bx[8 * a + k] = +1.0f; bx[8 * a + 4 + k] = -1.0f; }   // never hit (rtk.c:1612-1620)
				ptr[k] = b->leaf_off | 1u;
				continue;
			}
			const Node &n = b->nodes[(size_t)c[k]];
			for (int a = 0; a < 3; a++) { bx[8 * a + k] = n.mn[a]; bx[8 * a + 4 + k] = n.mx[a]; }
			ptr[k] = ref >= 0 ? b->node_off + (uint64_t)ref * 128u : ((b->leaf_off + b->leaf_offset[(size_t)(-3 - ref)]) | 1u);
		}
	}
	for (size_t l = 0; l < b->leaf_nodes.size(); l++) {
		const Node &n = b->nodes[(size_t)b->leaf_nodes[l]];
		char *dst = blob + b->leaf_off + b->leaf_offset[l];
		const Group *g = b->groups[(size_t)n.group];
		const uint64_t info = (uint64_t)n.count | (b->vert_off + g->byte_offset);
		memcpy(dst, &info, 8);
		uint32_t *table = (uint32_t *)(dst + 8 + 8 * ((n.count + 3) & ~(size_t)3));
		uint32_t nm = 0;
		// triangles of a leaf in (mesh, triangle) order: the tie rule of the trace path is "first in leaf order"
		std::vector<size_t> order(n.count);
		for (size_t i = 0; i < n.count; i++) order[i] = n.begin + i;
		std::sort(order.begin(), order.end(), [&](size_t x, size_t y) {
			const Item &p = b->items[x], &q = b->items[y];
			return p.mesh != q.mesh ? p.mesh < q.mesh : p.tri < q.tri;
		});
		for (size_t i = 0; i < n.count; i++) {
			const Item &it = b->items[order[i]];
			uint8_t *rec = (uint8_t *)dst + 8 + 8 * i;
			rec[0] = it.local[0]; rec[1] = it.local[1]; rec[2] = it.local[2];
			uint32_t k = 0;
			for (; k < nm; k++) if (table[k] == it.mesh) break;
			if (k == nm) table[nm++] = it.mesh;
			rec[3] = (uint8_t)k;
			memcpy(rec + 4, &it.tri, 4);
		}
	}
	for (const Group *g : b->groups)
		if (!g->verts.empty()) memcpy(blob + b->vert_off + g->byte_offset, g->verts.data(), g->verts.size() * sizeof(rtk_vertex));
	return true;
}

void rtk_cpu_build_free(rtk_cpu_build *b)
{
	if (!b) return;
	for (Group *g : b->groups) delete g;
	delete b;
}

void rtk_cpu_build_stats(const rtk_cpu_build *b, size_t *wide_nodes, size_t *leaves, size_t *groups, size_t *vertices)
{
	size_t v = 0;
	for (const Group *g : b->groups) v += g->verts.size();
	if (wide_nodes) *wide_nodes = b->wide_root.size();
	if (leaves) *leaves = b->leaf_nodes.size();
	if (groups) *groups = b->groups.size();
	if (vertices) *vertices = v;
}
