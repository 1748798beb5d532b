// BVH quality lab (CPU, throw-away): binary builders (LBVH, PLOC) -> SAH leaf decision -> greedy BVH4 collapse
// -> ordered closest-hit traversal counting node/leaf/triangle visits per ray.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>

struct Box { float mn[3], mx[3]; };
static inline Box merge(const Box &a, const Box &b) { Box r; for (int k = 0; k < 3; k++) { r.mn[k] = std::min(a.mn[k], b.mn[k]); r.mx[k] = std::max(a.mx[k], b.mx[k]); } return r; }
static inline float harea(const Box &b) { float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2]; return x * y + y * z + z * x; }

struct Ray { float o[3], d[3], tmin, tmax; };

static int KEYSHIFT = 0;
static std::vector<float> tris;   // 9 per tri
static size_t N;

static uint64_t spread21(uint32_t v) {
	uint64_t x = v & 0x1fffffu;
	x = (x | (x << 32)) & 0x1f00000000ffffull; x = (x | (x << 16)) & 0x1f0000ff0000ffull; x = (x | (x << 8)) & 0x100f00f00f00f00full;
	x = (x | (x << 4)) & 0x10c30c30c30c30c3ull; x = (x | (x << 2)) & 0x1249249249249249ull; return x;
}

// binary tree: nodes 0..n-2 inner; child >=0 inner, <0 leaf ~sorted_index
struct Bin { int l, r; Box b; uint32_t cnt; float cost; bool leaf; int lo, hi; };
static std::vector<Bin> bin; static int root;
static std::vector<uint32_t> order;  // sorted index -> prim
static std::vector<Box> pbox;        // per sorted index
static std::vector<uint64_t> keys;

static void morton_sort() {
	std::vector<float> c2(3 * N);
	float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
	std::vector<Box> pb(N);
	for (size_t i = 0; i < N; i++) {
		const float *p = &tris[9 * i];
		for (int a = 0; a < 3; a++) {
			float mn = std::min(std::min(p[a], p[3 + a]), p[6 + a]), mx = std::max(std::max(p[a], p[3 + a]), p[6 + a]);
			pb[i].mn[a] = mn; pb[i].mx[a] = mx; c2[3 * i + a] = mn + mx; lo[a] = std::min(lo[a], mn + mx); hi[a] = std::max(hi[a], mn + mx);
		}
	}
	std::vector<std::pair<uint64_t, uint32_t>> kv(N);
	for (size_t i = 0; i < N; i++) {
		uint32_t q[3];
		for (int a = 0; a < 3; a++) { float ext = hi[a] - lo[a]; float t = ext > 0 ? (c2[3 * i + a] - lo[a]) / ext : 0; t = std::min(std::max(t, 0.f), 1.f); uint32_t v = (uint32_t)(t * 2097152.0f); q[a] = v > 2097151u ? 2097151u : v; }
		kv[i] = { ((spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2])) >> KEYSHIFT, (uint32_t)i };
	}
	std::stable_sort(kv.begin(), kv.end(), [](auto &a, auto &b) { return a.first < b.first; });
	order.resize(N); pbox.resize(N); keys.resize(N);
	for (size_t i = 0; i < N; i++) { order[i] = kv[i].second; pbox[i] = pb[kv[i].second]; keys[i] = kv[i].first; }
}

// ---- LBVH (same topology as Karras): recursive split at highest differing bit; equal keys split by index bits
static int lbvh_rec(int lo, int hi) {   // returns child ref
	if (lo == hi) return ~lo;
	int split;
	uint64_t a = keys[lo], b = keys[hi];
	if (a == b) {
		// karras uses index as tiebreak: delta = 64 + clz(i^j)
		int pre = __builtin_clz((unsigned)(lo ^ hi));
		// find largest s in [lo,hi) with clz(lo ^ s) > pre
		int s = lo; int step = hi - lo;
		do { step = (step + 1) >> 1; int ns = s + step; if (ns < hi && __builtin_clz((unsigned)(lo ^ ns)) > pre) s = ns; } while (step > 1);
		split = s;
	} else {
		int pre = __builtin_clzll(a ^ b);
		int s = lo; int step = hi - lo;
		do { step = (step + 1) >> 1; int ns = s + step; if (ns < hi) { uint64_t k = keys[ns]; int d = (k == a) ? 64 + __builtin_clz((unsigned)(lo ^ ns)) : __builtin_clzll(a ^ k); if (d > pre) s = ns; } } while (step > 1);
		split = s;
	}
	int id = (int)bin.size(); bin.push_back(Bin());
	int l = lbvh_rec(lo, split), r = lbvh_rec(split + 1, hi);
	bin[id].l = l; bin[id].r = r; bin[id].lo = lo; bin[id].hi = hi;
	return id;
}

static void refit_rec(int id) {
	Bin &n = bin[id];
	Box b[2]; uint32_t c[2];
	for (int s = 0; s < 2; s++) { int ch = s ? n.r : n.l; if (ch < 0) { b[s] = pbox[~ch]; c[s] = 1; } else { refit_rec(ch); b[s] = bin[ch].b; c[s] = bin[ch].cnt; } }
	n.b = merge(b[0], b[1]); n.cnt = c[0] + c[1];
}

// ---- PLOC
static void build_ploc(int R) {
	bin.clear(); bin.reserve(N);
	struct Cl { Box b; int ref; };
	std::vector<Cl> C(N), C2;
	for (size_t i = 0; i < N; i++) { C[i].b = pbox[i]; C[i].ref = ~(int)i; }
	std::vector<int> nn;
	int iters = 0;
	while (C.size() > 1) {
		int m = (int)C.size();
		nn.assign(m, -1);
		for (int i = 0; i < m; i++) {
			float best = INFINITY; int bj = -1;
			int j0 = std::max(0, i - R), j1 = std::min(m - 1, i + R);
			for (int j = j0; j <= j1; j++) { if (j == i) continue; float a = harea(merge(C[i].b, C[j].b)); if (a < best) { best = a; bj = j; } }
			nn[i] = bj;
		}
		C2.clear();
		for (int i = 0; i < m; i++) {
			int j = nn[i];
			if (nn[j] == i) {
				if (i < j) {
					Bin b; b.l = C[i].ref; b.r = C[j].ref; b.b = merge(C[i].b, C[j].b);
					uint32_t cl = b.l < 0 ? 1 : bin[b.l].cnt, cr = b.r < 0 ? 1 : bin[b.r].cnt; b.cnt = cl + cr; b.cost = 0; b.leaf = false;
					int id = (int)bin.size(); bin.push_back(b);
					C2.push_back({ b.b, id });
				}
			} else C2.push_back(C[i]);
		}
		C.swap(C2); iters++;
	}
	root = C[0].ref;
	fprintf(stderr, "ploc R=%d iterations %d nodes %zu\n", R, iters, bin.size());
}


// ---- binned SAH top-down over an arbitrary list of sorted indices (idx values index pbox/order)
static int BIN_OVER_BOX = 0;
static int SAH_BINS = 32; static int SAH_LEAF = 1;
static int sah_build(std::vector<int> &idx, int lo, int hi) {   // [lo,hi)
	int n = hi - lo;
	if (n == 1) return ~idx[lo];
	Box cb; for (int k = 0; k < 3; k++) { cb.mn[k] = 1e30f; cb.mx[k] = -1e30f; }
	Box nb = cb;
	for (int i = lo; i < hi; i++) { const Box &b = pbox[idx[i]]; nb = merge(nb, b); for (int k = 0; k < 3; k++) { float c = b.mn[k] + b.mx[k]; cb.mn[k] = std::min(cb.mn[k], c); cb.mx[k] = std::max(cb.mx[k], c); } }
	if (BIN_OVER_BOX) { for (int k = 0; k < 3; k++) { cb.mn[k] = 2 * nb.mn[k]; cb.mx[k] = 2 * nb.mx[k]; } }   // bin over the node's box instead of its centroid bounds
	float bestc = INFINITY; int besta = -1, bestb = -1;
	const int NB = SAH_BINS < 0 ? std::min(-SAH_BINS, n) : SAH_BINS;      // negative: at most that many, never more bins than items
	for (int a = 0; a < 3; a++) {
		float ext = cb.mx[a] - cb.mn[a]; if (!(ext > 0)) continue;
		std::vector<Box> bb(NB); std::vector<int> bc(NB, 0);
		for (int b = 0; b < NB; b++) for (int k = 0; k < 3; k++) { bb[b].mn[k] = 1e30f; bb[b].mx[k] = -1e30f; }
		float sc = NB / ext;
		for (int i = lo; i < hi; i++) { const Box &b = pbox[idx[i]]; int bi = std::min(NB - 1, (int)(((b.mn[a] + b.mx[a]) - cb.mn[a]) * sc)); bb[bi] = merge(bb[bi], b); bc[bi]++; }
		std::vector<float> ra(NB); std::vector<int> rc(NB);
		Box acc; for (int k = 0; k < 3; k++) { acc.mn[k] = 1e30f; acc.mx[k] = -1e30f; } int cnt = 0;
		for (int b = NB - 1; b >= 1; b--) { if (bc[b]) acc = merge(acc, bb[b]); cnt += bc[b]; ra[b] = cnt ? harea(acc) : 0; rc[b] = cnt; }
		for (int k = 0; k < 3; k++) { acc.mn[k] = 1e30f; acc.mx[k] = -1e30f; } cnt = 0;
		for (int b = 0; b < NB - 1; b++) { if (bc[b]) acc = merge(acc, bb[b]); cnt += bc[b]; if (cnt == 0 || rc[b + 1] == 0) continue; float c = harea(acc) * cnt + ra[b + 1] * rc[b + 1]; if (c < bestc) { bestc = c; besta = a; bestb = b; } }
	}
	int mid;
	if (besta < 0) { mid = lo + n / 2; }
	else {
		float ext = cb.mx[besta] - cb.mn[besta], sc = NB / ext;
		auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](int i) { const Box &b = pbox[i]; int bi = std::min(NB - 1, (int)(((b.mn[besta] + b.mx[besta]) - cb.mn[besta]) * sc)); return bi <= bestb; });
		mid = (int)(it - idx.begin());
		if (mid == lo || mid == hi) mid = lo + n / 2;
	}
	int id = (int)bin.size(); bin.push_back(Bin());
	int l = sah_build(idx, lo, mid), r = sah_build(idx, mid, hi);
	bin[id].l = l; bin[id].r = r;
	return id;
}
// LBVH above, SAH rebuild of every maximal subtree with <= T prims
static int HYB_T = 256;
static int hyb_rec(int lo, int hi) {
	if (hi - lo + 1 <= HYB_T) { std::vector<int> idx(hi - lo + 1); for (int i = lo; i <= hi; i++) idx[i - lo] = i; return sah_build(idx, 0, (int)idx.size()); }
	int split; uint64_t a = keys[lo], b = keys[hi];
	if (a == b) split = (lo + hi) / 2;
	else { int pre = __builtin_clzll(a ^ b); int s = lo; int step = hi - lo; do { step = (step + 1) >> 1; int ns = s + step; if (ns < hi) { uint64_t k = keys[ns]; int d = (k == a) ? 64 : __builtin_clzll(a ^ k); if (d > pre) s = ns; } } while (step > 1); split = s; }
	int id = (int)bin.size(); bin.push_back(Bin());
	int l = hyb_rec(lo, split), r = hyb_rec(split + 1, hi);
	bin[id].l = l; bin[id].r = r; return id;
}

// LBVH above; below T prims the split position along the MORTON ORDER is chosen by the surface-area heuristic
// (prefix / suffix box sweep): triangles stay where the sort put them, only the topology changes
static int swp_build(int lo, int hi) {   // inclusive sorted range
	if (lo == hi) return ~lo;
	int n = hi - lo + 1;
	std::vector<float> suf(n);
	Box acc = pbox[hi]; suf[n - 1] = harea(acc);
	for (int i = hi - 1; i > lo; i--) { acc = merge(acc, pbox[i]); suf[i - lo] = harea(acc); }
	acc = pbox[lo]; float bestc = INFINITY; int best = lo;
	for (int s = lo; s < hi; s++) {
		if (s > lo) acc = merge(acc, pbox[s]);
		float c = harea(acc) * (float)(s - lo + 1) + suf[s + 1 - lo] * (float)(hi - s);
		if (c < bestc) { bestc = c; best = s; }
	}
	int id = (int)bin.size(); bin.push_back(Bin());
	int l = swp_build(lo, best), r = swp_build(best + 1, hi);
	bin[id].l = l; bin[id].r = r; return id;
}
static int mswp_rec(int lo, int hi) {
	if (hi - lo + 1 <= HYB_T) return swp_build(lo, hi);
	int split; uint64_t a = keys[lo], b = keys[hi];
	if (a == b) split = (lo + hi) / 2;
	else { int pre = __builtin_clzll(a ^ b); int s = lo; int step = hi - lo; do { step = (step + 1) >> 1; int ns = s + step; if (ns < hi) { uint64_t k = keys[ns]; int d = (k == a) ? 64 : __builtin_clzll(a ^ k); if (d > pre) s = ns; } } while (step > 1); split = s; }
	int id = (int)bin.size(); bin.push_back(Bin());
	int l = mswp_rec(lo, split), r = mswp_rec(split + 1, hi);
	bin[id].l = l; bin[id].r = r; return id;
}

// fixed groups of T consecutive sorted prims rebuilt by SAH; above them the radix tree over the GROUP borders only
// (what a bottom-up Apetrei climb gives when it starts from the groups instead of the leaves)
static int grp_rec(int g0, int g1) {   // inclusive group range
	if (g0 == g1) { int lo = g0 * HYB_T, hi = std::min((int)N, lo + HYB_T); std::vector<int> idx(hi - lo); for (int i = lo; i < hi; i++) idx[i - lo] = i; return sah_build(idx, 0, (int)idx.size()); }
	// border after group g: similarity of keys[(g+1)*T-1] and keys[(g+1)*T]; split at the least similar border (first one on ties of equal keys -> middle)
	int best = g0; int bestd = 1 << 30;
	for (int g = g0; g < g1; g++) { uint64_t a = keys[(size_t)(g + 1) * HYB_T - 1], b = keys[(size_t)(g + 1) * HYB_T]; int d = a == b ? 64 + __builtin_clz((unsigned)(((g + 1) * HYB_T - 1) ^ ((g + 1) * HYB_T))) : __builtin_clzll(a ^ b); if (d < bestd) { bestd = d; best = g; } }
	int id = (int)bin.size(); bin.push_back(Bin());
	int l = grp_rec(g0, best), r = grp_rec(best + 1, g1);
	bin[id].l = l; bin[id].r = r; return id;
}

// ---- SAH leaf decision (bottom-up): order-independent recursion
static float CT = 1.0f, CN = 0.5f; static uint32_t MAXLEAF = 8;
static float sah_rec(int id) {
	Bin &n = bin[id];
	float cost = 0;
	for (int s = 0; s < 2; s++) { int ch = s ? n.r : n.l; if (ch < 0) cost += CT * harea(pbox[~ch]); else cost += sah_rec(ch); }
	float area = harea(n.b);
	float split = CN * area + cost;
	float leaf = n.cnt <= MAXLEAF ? CT * (float)n.cnt * area : INFINITY;
	n.leaf = leaf <= split; n.cost = std::min(leaf, split);
	return n.cost;
}

// ---- BVH4
static int WIDTH = 4; static int ORDERMODE = 0;
struct W { Box b[8]; int ref[8]; /* >=0 wide idx; <0: leaf id ~k */ int n; };
struct Leaf { std::vector<uint32_t> prims; };
static std::vector<W> wide; static std::vector<Leaf> leaves;
static int COLLAPSE_CRIT = 0;  // which open child the greedy collapse opens next: 0 largest area, 1 area x count, 2 area saved, 3 count
static int COLLAPSE_MODE = 0;   // 0 greedy largest area; 1 fixed two-level

static void gather_prims(int ref, std::vector<uint32_t> &out) { if (ref < 0) { out.push_back(order[~ref]); return; } gather_prims(bin[ref].l, out); gather_prims(bin[ref].r, out); }

static int make_leaf(int ref) { Leaf l; gather_prims(ref, l.prims); leaves.push_back(l); return ~(int)(leaves.size() - 1); }

static Box ref_box(int ref) { return ref < 0 ? pbox[~ref] : bin[ref].b; }
static int FORCE_TILE = 0;   // > 0: a binary node whose sorted range lies inside one tile of this many triangles is never opened INSIDE a wide node above it: it becomes a wide node of its own (the device build's tile-local collapse)
static bool ref_open(int ref) { return ref >= 0 && !bin[ref].leaf; }
static bool in_tile(int ref) { return FORCE_TILE > 0 && bin[ref].lo / FORCE_TILE == bin[ref].hi / FORCE_TILE; }

static void collapse() {
	wide.clear(); leaves.clear();
	std::vector<std::pair<int, int>> q;  // (bin, wide idx)
	wide.push_back(W()); q.push_back({ root, 0 });
	for (size_t qi = 0; qi < q.size(); qi++) {
		int b = q[qi].first, wi = q[qi].second;
		int c[8]; int nc;
		if (bin[b].leaf) { c[0] = b; nc = 1; }
		else {
			c[0] = bin[b].l; c[1] = bin[b].r; nc = 2;
			if (COLLAPSE_MODE == 0) {
				for (int round = 0; round < WIDTH - 2; round++) {
					int best = -1; float ba = 0;
					for (int k = 0; k < nc; k++) if (ref_open(c[k]) && !(in_tile(c[k]) && !in_tile(b))) {
						float a = harea(bin[c[k]].b);
						if (COLLAPSE_CRIT == 1) a *= (float)bin[c[k]].cnt;                                   // area x triangles below
						else if (COLLAPSE_CRIT == 2) a = a - 0.5f * (harea(ref_box(bin[c[k]].l)) + harea(ref_box(bin[c[k]].r)));   // what opening saves
						else if (COLLAPSE_CRIT == 3) a = (float)bin[c[k]].cnt;                               // most triangles below
						if (a > ba) { ba = a; best = k; } }
					if (best < 0) break;
					int o = c[best]; c[best] = bin[o].l; c[nc++] = bin[o].r;
				}
			} else {
				int c2[4]; int n2 = 0;
				for (int k = 0; k < 2; k++) { if (ref_open(c[k])) { c2[n2++] = bin[c[k]].l; c2[n2++] = bin[c[k]].r; } else c2[n2++] = c[k]; }
				nc = n2; for (int k = 0; k < nc; k++) c[k] = c2[k];
			}
		}
		W w; w.n = nc;
		for (int k = 0; k < nc; k++) {
			w.b[k] = ref_box(c[k]);
			if (ref_open(c[k])) { int idx = (int)wide.size(); wide.push_back(W()); q.push_back({ c[k], idx }); w.ref[k] = idx; }
			else w.ref[k] = make_leaf(c[k]);
		}
		wide[wi] = w;
	}
}

// ---- traversal with counters
struct Cnt { uint64_t nodes = 0, leaves = 0, tris = 0, hits = 0; uint64_t sp_hist[40] = {0}; uint64_t max_hist[40] = {0}; };
static bool tri_hit(const Ray &r, const float *p, double &t) {
	double e1[3], e2[3], pv[3], tv[3], qv[3];
	for (int k = 0; k < 3; k++) { e1[k] = (double)p[3 + k] - p[k]; e2[k] = (double)p[6 + k] - p[k]; }
	pv[0] = r.d[1] * e2[2] - r.d[2] * e2[1]; pv[1] = r.d[2] * e2[0] - r.d[0] * e2[2]; pv[2] = r.d[0] * e2[1] - r.d[1] * e2[0];
	double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
	if (det == 0) return false;
	double inv = 1.0 / det;
	for (int k = 0; k < 3; k++) tv[k] = (double)r.o[k] - p[k];
	double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv; if (u < 0 || u > 1) return false;
	qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
	double v = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) * inv; if (v < 0 || u + v > 1) return false;
	t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
	return true;
}

static void trace(const Ray &r, Cnt &c) {
	float best = r.tmax; bool hit = false;
	float rd[3] = { 1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2] };
	struct E { float t; int ref; } stack[256]; int sp = 0;
	int top = 0; float topt = 0; int maxsp = 0;
	for (;;) {
		if (top >= 0) {
			const W &w = wide[top]; c.nodes++;
			float key[8]; int ref[8]; int nh = 0;
			for (int k = 0; k < w.n; k++) {
				float tn = r.tmin, tf = best;
				for (int a = 0; a < 3; a++) { float t0 = (w.b[k].mn[a] - r.o[a]) * rd[a], t1 = (w.b[k].mx[a] - r.o[a]) * rd[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
				if (tn <= tf) { key[nh] = tn; ref[nh] = w.ref[k]; nh++; }
			}
			if (ORDERMODE == 0) { for (int i = 1; i < nh; i++) for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); } }
			else if (nh > 1) { int m = 0; for (int i = 1; i < nh; i++) if (key[i] < key[m]) m = i; std::swap(key[0], key[m]); std::swap(ref[0], ref[m]); }   // nearest first, rest in slot order
			for (int i = nh - 1; i >= 1; i--) { stack[sp].t = key[i]; stack[sp].ref = ref[i]; sp++; }
			c.sp_hist[std::min(sp, 39)]++; maxsp = std::max(maxsp, sp);
			if (nh) { top = ref[0]; topt = key[0]; continue; }
		} else {
			const Leaf &l = leaves[~top]; c.leaves++;
			for (uint32_t p : l.prims) { c.tris++; double t; if (tri_hit(r, &tris[9 * (size_t)p], t) && t > r.tmin && t < best) { best = (float)t; hit = true; } }
		}
		bool got = false;
		while (sp > 0) { sp--; if (stack[sp].t > best) continue; top = stack[sp].ref; got = true; break; }
		if (!got) break;
	}
	(void)topt; c.max_hist[std::min(maxsp, 39)]++;
	if (hit) c.hits++;
}


// ---- packet simulation: the control flow of rtk_trace_packet_kernel (one 8x8 tile per wave) on the CPU, counting wave-level steps
// by kind, so that changes of the traversal's SHAPE (entry points shared by a 64x64 block, child order, packet size) can be
// priced before they are written for the GPU.
struct PkCnt { uint64_t tiles = 0, steps[5] = {0,0,0,0,0}, pops = 0, pops_culled = 0, tri_steps = 0, pushes = 0, lane_nodes = 0, entries = 0, entry_culled = 0, pre_steps = 0, blocks = 0, entry_list = 0; };
static int PK_ANY = 0;   // any-hit packets: a lane that found a hit takes no further part
static int PK_BLOCKS = 0; static int PK_LANES = 64; static int PK_ENTRY_DEPTH = 0; static int PK_ORDER = 0; static int PK_ENTRY_MAX = 1 << 30;
static std::vector<int> wdepth;   // depth of every wide node (root = 0)
static std::vector<uint8_t> wperm;  // per node, per octant: child order (4 x 2 bits), by distance of the child box centre along the octant's diagonal

static inline bool slab(const Ray &r, const float *rd, const Box &b, float best, float &tn) {
	float n = r.tmin, f = best;
	for (int a = 0; a < 3; a++) { float t0 = (b.mn[a] - r.o[a]) * rd[a], t1 = (b.mx[a] - r.o[a]) * rd[a]; if (t0 > t1) std::swap(t0, t1); n = std::max(n, t0); f = std::min(f, t1); }
	tn = n; return n <= f;
}

struct PkEntry { int ref; float tlo; };

// conservative interval slab test of a whole block of rays (same direction signs): lower bound of the entry distance
static bool block_slab(const std::vector<Ray> &rs, const Box &b, float &tlo) {
	float lo = 1e30f; bool any = false;
	for (const Ray &r : rs) { float rd[3] = { 1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2] }; float tn; if (slab(r, rd, b, r.tmax, tn)) { any = true; lo = std::min(lo, tn); } }
	tlo = lo; return any;
}

// What a GPU pre-pass can actually compute (PK_BEAM = 1): the block's rays bounded by an origin box and a box of reciprocal
// directions (same signs), and the interval form of the slab test -- per axis a lower bound of the entry parameter and an upper
// bound of the exit parameter over ALL rays of the beam. A superset of the exact union above.
static int PK_BEAM = 0;
static int PK_TILEBEAM = 0;
static int PK_CACHE_N = 0, PK_CACHE_T = 0, PK_ZORDER = 0;   // software cache of nodes / triangles per workgroup (direct-mapped slots); tiles of a block in Z order
struct SwCache { std::vector<int> ntag, ttag; uint64_t nh = 0, nm = 0, th = 0, tm = 0; };
static thread_local SwCache *g_cache = nullptr;   // the node test of a tile is the interval test of the TILE's own beam (wave-uniform), not 64 per-lane slab tests
struct Beam { float olo[3], ohi[3], rlo[3], rhi[3]; int neg[3]; float tmin, tmax; };
static Beam make_beam(const std::vector<Ray> &rs) {
	Beam b; for (int a = 0; a < 3; a++) { b.olo[a] = 1e30f; b.ohi[a] = -1e30f; b.rlo[a] = 1e30f; b.rhi[a] = -1e30f; b.neg[a] = rs[0].d[a] < 0; }
	b.tmin = 1e30f; b.tmax = -1e30f;
	for (const Ray &r : rs) { for (int a = 0; a < 3; a++) { const float rd = 1.0f / r.d[a]; b.olo[a] = std::min(b.olo[a], r.o[a]); b.ohi[a] = std::max(b.ohi[a], r.o[a]); b.rlo[a] = std::min(b.rlo[a], rd); b.rhi[a] = std::max(b.rhi[a], rd); } b.tmin = std::min(b.tmin, r.tmin); b.tmax = std::max(b.tmax, r.tmax); }
	return b;
}
static bool beam_slab(const Beam &b, const Box &bx, float &tlo) {
	float n = b.tmin, f = b.tmax;
	for (int a = 0; a < 3; a++) {
		// entry plane p_n, exit plane p_f by direction sign; t = (p - o) * rd over o in [olo, ohi], rd in [rlo, rhi] (one sign)
		const float pn = b.neg[a] ? bx.mx[a] : bx.mn[a], pf = b.neg[a] ? bx.mn[a] : bx.mx[a];
		float lo = 1e30f, hi = -1e30f;
		for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) { const float o = i ? b.ohi[a] : b.olo[a], rd = j ? b.rhi[a] : b.rlo[a]; lo = std::min(lo, (pn - o) * rd); hi = std::max(hi, (pf - o) * rd); }
		n = std::max(n, lo); f = std::min(f, hi);
	}
	tlo = n; return n <= f;
}

static void block_entries(const std::vector<Ray> &rs, std::vector<PkEntry> &out, PkCnt &c) {
	if (PK_BEAM) {
		const Beam bm = make_beam(rs);
		out.clear();
		struct S { int ref; float tlo; }; std::vector<S> st; st.push_back({ 0, 0.f });
		while (!st.empty()) {
			S s = st.back(); st.pop_back();
			if (s.ref < 0 || wdepth[s.ref] >= PK_ENTRY_DEPTH) { out.push_back({ s.ref, s.tlo }); continue; }
			c.pre_steps++;
			const W &w = wide[s.ref];
			for (int k = 0; k < w.n; k++) { float tlo; if (beam_slab(bm, w.b[k], tlo)) st.push_back({ w.ref[k], tlo }); }
		}
		std::sort(out.begin(), out.end(), [](const PkEntry &a, const PkEntry &b) { return a.tlo < b.tlo; });
		return;
	}
	// exact union over the block's rays of "enters this node" down to depth PK_ENTRY_DEPTH (what a conservative pre-pass approximates from above)
	out.clear();
	struct S { int ref; float tlo; }; std::vector<S> st; st.push_back({ 0, 0.f });
	while (!st.empty()) {
		S s = st.back(); st.pop_back();
		if (s.ref < 0 || wdepth[s.ref] >= PK_ENTRY_DEPTH) { out.push_back({ s.ref, s.tlo }); continue; }
		c.pre_steps++;
		const W &w = wide[s.ref];
		for (int k = 0; k < w.n; k++) { float tlo; if (block_slab(rs, w.b[k], tlo)) st.push_back({ w.ref[k], tlo }); }
	}
	std::sort(out.begin(), out.end(), [](const PkEntry &a, const PkEntry &b) { return a.tlo < b.tlo; });
}

static void trace_packet(const std::vector<Ray> &rs, PkCnt &c, const std::vector<PkEntry> *entries, std::vector<float> *tout) {
	const int L = (int)rs.size();
	std::vector<float> best(L), rd(3 * L);
	std::vector<uint8_t> hit(L, 0);
	for (int i = 0; i < L; i++) { best[i] = rs[i].tmax; for (int a = 0; a < 3; a++) rd[3 * i + a] = 1.0f / rs[i].d[a]; }
	struct E { int ref; std::vector<float> te; };
	std::vector<E> stack;
	std::vector<uint8_t> live(L, 1);
	int top = 0; size_t next_entry = 0;
	bool need_entry = entries != nullptr;
	c.tiles++;
	const int oct = (rs[0].d[0] < 0 ? 1 : 0) | (rs[0].d[1] < 0 ? 2 : 0) | (rs[0].d[2] < 0 ? 4 : 0);
	for (;;) {
		bool pop = false;
		if (need_entry) {
			// take the next entry point of the block that some lane can still reach
			bool got = false;
			while (next_entry < entries->size()) {
				const PkEntry &e = (*entries)[next_entry++]; c.entries++;
				bool any = false; for (int i = 0; i < L; i++) { live[i] = e.tlo <= best[i]; any |= live[i]; }
				if (!any) { c.entry_culled++; next_entry = entries->size(); break; }   // sorted by lower bound: nothing behind it can matter either
				top = e.ref; got = true; break;
			}
			if (!got) break;
			need_entry = false;
		}
		if (top >= 0) {
			const W &w = wide[top];
			if (g_cache && PK_CACHE_N) { int &tg = g_cache->ntag[(size_t)top % PK_CACHE_N]; if (tg == top) g_cache->nh++; else { g_cache->nm++; tg = top; } }
			float pay[8][256]; bool any[8]; int n_any = 0;
			for (int i = 0; i < L; i++) if (live[i]) c.lane_nodes++;
			if (PK_TILEBEAM) {
				Beam tb = make_beam(rs); tb.tmax = -1e30f; for (int i = 0; i < L; i++) tb.tmax = std::max(tb.tmax, best[i]);
				for (int k = 0; k < w.n; k++) { float tlo; any[k] = beam_slab(tb, w.b[k], tlo); for (int i = 0; i < L; i++) pay[k][i] = any[k] ? tlo : NAN; n_any += any[k]; }
			} else
			for (int k = 0; k < w.n; k++) {
				any[k] = false;
				for (int i = 0; i < L; i++) { float tn; if (live[i] && slab(rs[i], &rd[3 * i], w.b[k], best[i], tn)) { pay[k][i] = tn; any[k] = true; } else pay[k][i] = NAN; }
				n_any += any[k];
			}
			c.steps[std::min(n_any, 4)]++;
			if (n_any == 0) pop = true;
			else {
				int lead = 0; while (!live[lead]) lead++;
				int idx[8], m = 0;
				for (int k = 0; k < w.n; k++) if (any[k]) idx[m++] = k;
				if (PK_ORDER == 0) {
					auto key = [&](int k) { float p = pay[k][lead]; return std::isnan(p) ? 3e38f : p; };
					std::stable_sort(idx, idx + m, [&](int a, int b) { return key(a) < key(b); });
				} else {
					const uint8_t pm = wperm[(size_t)top * 8 + oct]; int pos[4]; for (int q = 0; q < 4; q++) pos[(pm >> (2 * q)) & 3] = q;
					std::stable_sort(idx, idx + m, [&](int a, int b) { return pos[a] < pos[b]; });
				}
				for (int q = m - 1; q >= 1; q--) { E e; e.ref = w.ref[idx[q]]; e.te.assign(pay[idx[q]], pay[idx[q]] + L); stack.push_back(std::move(e)); c.pushes++; }
				for (int i = 0; i < L; i++) live[i] = !std::isnan(pay[idx[0]][i]);
				top = w.ref[idx[0]];
			}
		} else {
			const Leaf &l = leaves[~top];
			for (uint32_t p : l.prims) {
				c.tri_steps++;
				if (g_cache && PK_CACHE_T) { int &tg = g_cache->ttag[(size_t)p % PK_CACHE_T]; if (tg == (int)p) g_cache->th++; else { g_cache->tm++; tg = (int)p; } }
				for (int i = 0; i < L; i++) if (live[i]) { double t; if (tri_hit(rs[i], &tris[9 * (size_t)p], t) && t > rs[i].tmin && t < best[i]) { best[i] = PK_ANY ? -3e38f : (float)t; hit[i] = 1; } }
			}
			pop = true;
		}
		if (pop) {
			bool found = false;
			while (!stack.empty()) {
				E e = std::move(stack.back()); stack.pop_back(); c.pops++;
				bool any = false; for (int i = 0; i < L; i++) { live[i] = e.te[i] <= best[i]; any |= live[i]; }   // NaN compares false
				if (any) { top = e.ref; found = true; break; }
				c.pops_culled++;
			}
			if (!found) { if (entries) need_entry = true; else break; }
		}
	}
	if (tout) *tout = best;
}

// PK_TILEBEAM == 2: the beam traversal with a wave-uniform stack of {ref, tlo}, testing up to TWO nodes per round (the
// current one and the top of the stack; lanes 0-31 / 32-63 of the plane-per-lane layout): rounds = dependent memory round trips
struct PairCnt { uint64_t tiles = 0, node_rounds = 0, nodes_tested = 0, tri_steps = 0, leaf_rounds = 0, pushes = 0, pops = 0, culled = 0; };
static void trace_packet_pair(const std::vector<Ray> &rs, PairCnt &c, const std::vector<PkEntry> *entries, std::vector<float> *tout, int width) {
	const int L = (int)rs.size();
	std::vector<float> best(L); for (int i = 0; i < L; i++) best[i] = rs[i].tmax;
	Beam tb = make_beam(rs);
	const int oct = (rs[0].d[0] < 0 ? 1 : 0) | (rs[0].d[1] < 0 ? 2 : 0) | (rs[0].d[2] < 0 ? 4 : 0);
	struct E { int ref; float tlo; };
	std::vector<E> stack; size_t next_entry = 0;
	c.tiles++;
	auto tmax = [&]() { float m = -1e30f; for (int i = 0; i < L; i++) m = std::max(m, best[i]); return m; };
	if (!entries) stack.push_back({ 0, 0.f });
	for (;;) {
		if (stack.empty()) {
			if (!entries || next_entry >= entries->size()) break;
			const PkEntry &e = (*entries)[next_entry++];
			if (e.tlo > tmax()) break;
			stack.push_back({ e.ref, e.tlo });
		}
		// pop up to `width` live entries
		std::vector<E> cur;
		const float tm = tmax();
		const bool mixed = width >= 10;            // -tb 12: two entries of ANY kind per round (a leaf's triangle is fetched beside a node)
		const int wd = mixed ? width - 10 : width;
		while (!stack.empty() && (int)cur.size() < wd) {
			E e = stack.back();
			if (!mixed && e.ref < 0 && !cur.empty()) break;          // a leaf waits for its own round
			stack.pop_back(); c.pops++;
			if (e.tlo > tm) { c.culled++; continue; }
			cur.push_back(e);
			if (!mixed && e.ref < 0) break;
		}
		if (cur.empty()) continue;
		if (!mixed && cur[0].ref < 0) {
			const Leaf &l = leaves[~cur[0].ref];
			c.leaf_rounds++;
			for (uint32_t p : l.prims) { c.tri_steps++; for (int i = 0; i < L; i++) { double t; if (tri_hit(rs[i], &tris[9 * (size_t)p], t) && t > rs[i].tmin && t < best[i]) best[i] = (float)t; } }
			continue;
		}
		c.node_rounds++;
		tb.tmax = tm;
		// children of the LAST popped (deeper in the stack = farther) first, so that the first popped node's nearest child ends on top
		for (int q = (int)cur.size() - 1; q >= 0; q--) {
			if (cur[q].ref < 0) {
				const Leaf &l = leaves[~cur[q].ref];
				for (uint32_t p : l.prims) { c.tri_steps++; for (int i = 0; i < L; i++) { double t; if (tri_hit(rs[i], &tris[9 * (size_t)p], t) && t > rs[i].tmin && t < best[i]) best[i] = (float)t; } }
				continue;
			}
			c.nodes_tested++;
			const W &w = wide[cur[q].ref];
			int idx[8], m = 0; float tl[8];
			for (int k = 0; k < w.n; k++) { float tlo; if (beam_slab(tb, w.b[k], tlo)) { idx[m] = k; tl[k] = tlo; m++; } }
			const uint8_t pm = wperm[(size_t)cur[q].ref * 8 + oct]; int pos[4]; for (int z = 0; z < 4; z++) pos[(pm >> (2 * z)) & 3] = z;
			std::stable_sort(idx, idx + m, [&](int a, int b) { return pos[a] < pos[b]; });
			for (int z = m - 1; z >= 0; z--) { stack.push_back({ w.ref[idx[z]], tl[idx[z]] }); c.pushes++; }
		}
	}
	if (tout) *tout = best;
}

// PK_TILEBEAM == 20: one wave walks TWO adjacent 8x8 tiles (a 16x8-pixel tile as two ray groups with a beam each: lanes 0-31 / 32-63 of
// the plane-per-lane layout test the same node for the two groups): an entry carries the set of groups that enter it; a triangle is
// tested only for the groups whose own beam reaches its leaf. Counts per 128 rays: node steps, triangle tests per group.
struct GrpCnt { uint64_t tiles = 0, node_steps = 0, tri_group_tests = 0, leaves = 0, pushes = 0, pops = 0, culled = 0, t_geom = 0, t_acc = 0, t_edge = 0, t_edge_wrong = 0, t_far = 0; };
// Round 5: would a wave-uniform test of the triangle against the GROUP's beam have skipped this triangle test? Common origin: the edge
// function of edge k for a ray of direction d is n_k . d (n_k = p_(k+1) x p_(k+2), p = vertex - origin), linear in d, so its range over
// the group's direction box is the sum of the per-component extremes. Skipped if some edge function is negative for every ray of the
// box and another one positive for every ray (then the signs are mixed for every ray: rtk.c:340-344), with a relative margin.
static bool edge_cull(const std::vector<Ray> &g, const float *p) {
	double dlo[3] = { 1e30, 1e30, 1e30 }, dhi[3] = { -1e30, -1e30, -1e30 };
	for (const Ray &r : g) for (int a = 0; a < 3; a++) { dlo[a] = std::min(dlo[a], (double)r.d[a]); dhi[a] = std::max(dhi[a], (double)r.d[a]); }
	double q[3][3]; for (int v = 0; v < 3; v++) for (int a = 0; a < 3; a++) q[v][a] = (double)p[3 * v + a] - g[0].o[a];
	bool neg = false, pos = false;
	for (int k = 0; k < 3; k++) {
		const double *a = q[(k + 1) % 3], *b = q[(k + 2) % 3];
		const double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
		double mx = 0, mn = 0, mag = 0;
		for (int c = 0; c < 3; c++) { mx += std::max(n[c] * dlo[c], n[c] * dhi[c]); mn += std::min(n[c] * dlo[c], n[c] * dhi[c]); mag += std::max(std::fabs(n[c] * dlo[c]), std::fabs(n[c] * dhi[c])); }
		const double m = mag * (1.0 / 4096.0);
		if (mx < -m) neg = true;
		if (mn > m) pos = true;
	}
	return neg && pos;
}
static void trace_packet_groups(const std::vector<Ray> &rs, int tw, GrpCnt &c, const std::vector<PkEntry> *entries, std::vector<float> *tout) {
	const int L = (int)rs.size();
	std::vector<float> best(L); for (int i = 0; i < L; i++) best[i] = rs[i].tmax;
	std::vector<Ray> g[2]; std::vector<int> gi[2];
	for (int i = 0; i < L; i++) { const int grp = (i % tw) >= 8 ? 1 : 0; g[grp].push_back(rs[i]); gi[grp].push_back(i); }
	Beam tb[2] = { make_beam(g[0]), make_beam(g[1]) };
	const int oct = (rs[0].d[0] < 0 ? 1 : 0) | (rs[0].d[1] < 0 ? 2 : 0) | (rs[0].d[2] < 0 ? 4 : 0);
	struct E { int ref; float tlo; int gm; };
	std::vector<E> stack; size_t next_entry = 0;
	c.tiles++;
	auto tmax = [&](int grp) { float m = -1e30f; for (int i : gi[grp]) m = std::max(m, best[i]); return m; };
	if (!entries) stack.push_back({ 0, 0.f, 3 });
	for (;;) {
		if (stack.empty()) {
			if (!entries || next_entry >= entries->size()) break;
			const PkEntry &e = (*entries)[next_entry++];
			if (e.tlo > std::max(tmax(0), tmax(1))) break;
			stack.push_back({ e.ref, e.tlo, 3 });
		}
		E e = stack.back(); stack.pop_back(); c.pops++;
		const float tm[2] = { tmax(0), tmax(1) };
		int gm = e.gm; for (int q = 0; q < 2; q++) if (e.tlo > tm[q]) gm &= ~(1 << q);
		if (!gm) { c.culled++; continue; }
		if (e.ref < 0) {
			const Leaf &l = leaves[~e.ref]; c.leaves++;
			for (uint32_t p : l.prims) for (int q = 0; q < 2; q++) if (gm & (1 << q)) {
				c.tri_group_tests++;
				bool geom = false, acc = false; double tnear = 1e30;
				for (int i : gi[q]) { double t; if (tri_hit(rs[i], &tris[9 * (size_t)p], t)) { geom = true; tnear = std::min(tnear, t); if (t > rs[i].tmin && t < best[i]) { best[i] = (float)t; acc = true; } } }
				const bool ec = edge_cull(g[q], &tris[9 * (size_t)p]);
				c.t_geom += geom; c.t_acc += acc; c.t_edge += ec; c.t_edge_wrong += ec && geom; c.t_far += geom && !acc;
			}
			continue;
		}
		c.node_steps++;
		const W &w = wide[e.ref];
		int idx[8], m = 0; float tl[8]; int cg[8];
		for (int k = 0; k < w.n; k++) {
			int cm = 0; float lo = 1e30f;
			for (int q = 0; q < 2; q++) if (gm & (1 << q)) { tb[q].tmax = tm[q]; float tlo; if (beam_slab(tb[q], w.b[k], tlo)) { cm |= 1 << q; lo = std::min(lo, tlo); } }
			if (cm) { idx[m++] = k; tl[k] = lo; cg[k] = cm; }
		}
		const uint8_t pm = wperm[(size_t)e.ref * 8 + oct]; int pos[4]; for (int z = 0; z < 4; z++) pos[(pm >> (2 * z)) & 3] = z;
		std::stable_sort(idx, idx + m, [&](int a, int b) { return pos[a] < pos[b]; });
		for (int z = m - 1; z >= 0; z--) { stack.push_back({ w.ref[idx[z]], tl[idx[z]], cg[idx[z]] }); c.pushes++; }
	}
	if (tout) *tout = best;
}

static void packet_lab(int W_, int H_, int nblocks) {
	// depth + per-octant child order of every wide node
	wdepth.assign(wide.size(), 0);
	for (size_t i = 0; i < wide.size(); i++) for (int k = 0; k < wide[i].n; k++) if (wide[i].ref[k] >= 0) wdepth[wide[i].ref[k]] = wdepth[i] + 1;
	wperm.assign(wide.size() * 8, 0);
	for (size_t i = 0; i < wide.size(); i++) for (int o = 0; o < 8; o++) {
		const W &w = wide[i]; float key[4] = { 1e30f, 1e30f, 1e30f, 1e30f }; int idx[4] = { 0, 1, 2, 3 };
		for (int k = 0; k < w.n && k < 4; k++) { float s = 0; for (int a = 0; a < 3; a++) { float cen = PK_ORDER == 2 ? (((o >> a) & 1) ? 2 * w.b[k].mx[a] : 2 * w.b[k].mn[a]) : PK_ORDER == 3 ? (((o >> a) & 1) ? 2 * w.b[k].mn[a] : 2 * w.b[k].mx[a]) : w.b[k].mn[a] + w.b[k].mx[a]; s += ((o >> a) & 1) ? -cen : cen; } key[k] = s; }
		std::stable_sort(idx, idx + 4, [&](int a, int b) { return key[a] < key[b]; });
		uint8_t pm = 0; for (int q = 0; q < 4; q++) pm |= (uint8_t)(idx[q] << (2 * q));
		wperm[i * 8 + o] = pm;
	}
	const int tw = PK_LANES == 64 ? 8 : 16, th = PK_LANES == 256 ? 16 : 8;   // 8x8, 16x8, 16x16
	PkCnt c; PairCnt pc; GrpCnt gc; double tsum = 0; uint64_t mism = 0, gnh = 0, gnm = 0, gth = 0, gtm = 0;
	const int bx_n = W_ / 64, by_n = H_ / 64;
#pragma omp parallel
	{ PkCnt lc; PairCnt lpc; GrpCnt lgc; double ls = 0; uint64_t lm = 0, cnh = 0, cnm = 0, cth = 0, ctm = 0;
#pragma omp for schedule(dynamic, 1)
		for (int bi = 0; bi < nblocks; bi++) {
			const uint64_t h = (uint64_t)(bi + 1) * 0x9E3779B97F4A7C15ull;
			const int bx = (int)((h >> 20) % bx_n), by = (int)((h >> 40) % by_n);
			auto mkray = [&](int x, int y) { Ray r; r.o[0] = 0.5f; r.o[1] = 0.5f; r.o[2] = -1.5f; r.d[0] = ((x + 0.5f) / W_ - 0.5f) * 0.7f; r.d[1] = ((y + 0.5f) / H_ - 0.5f) * 0.7f; r.d[2] = 1.0f; r.tmin = 0; r.tmax = 3.402823e38f; return r; };
			std::vector<PkEntry> ent;
			if (PK_ENTRY_DEPTH > 0) {
				std::vector<Ray> all; for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++) all.push_back(mkray(bx * 64 + x, by * 64 + y));
				block_entries(all, ent, lc); lc.blocks++; lc.entry_list += ent.size();
			}
			SwCache cache; cache.ntag.assign(PK_CACHE_N ? PK_CACHE_N : 1, -1); cache.ttag.assign(PK_CACHE_T ? PK_CACHE_T : 1, -1); g_cache = &cache;
			for (int tz = 0; tz < (64 / th) * (64 / tw); tz++) {
				int ty = tz / (64 / tw), tx = tz % (64 / tw);
				if (PK_ZORDER && tw == 8 && th == 8) { tx = (tz & 1) | ((tz >> 1) & 2) | ((tz >> 2) & 4); ty = ((tz >> 1) & 1) | ((tz >> 2) & 2) | ((tz >> 3) & 4); }
				std::vector<Ray> rs; for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) rs.push_back(mkray(bx * 64 + tx * tw + x, by * 64 + ty * th + y));
				std::vector<float> t1; g_cache = &cache;
				if (PK_TILEBEAM == 20) { trace_packet_groups(rs, tw, lgc, PK_ENTRY_DEPTH > 0 ? &ent : nullptr, &t1); lc.tiles++; }
				else if (PK_TILEBEAM >= 2) { trace_packet_pair(rs, lpc, PK_ENTRY_DEPTH > 0 ? &ent : nullptr, &t1, PK_TILEBEAM); lc.tiles++; }
				else trace_packet(rs, lc, PK_ENTRY_DEPTH > 0 ? &ent : nullptr, &t1);
				for (size_t i = 0; i < rs.size(); i++) { Cnt cc; (void)cc; ls += t1[i] < 1e30f ? t1[i] : 0; }
				g_cache = nullptr;   // (the check run below is not part of the workgroup's work)
				if (PK_ENTRY_DEPTH > 0 || PK_ORDER) { PkCnt dummy; std::vector<float> t0; const int s0 = PK_ORDER; PK_ORDER = 0; trace_packet(rs, dummy, nullptr, &t0); PK_ORDER = s0; for (size_t i = 0; i < rs.size(); i++) lm += t0[i] != t1[i]; }
			}
			cnh += cache.nh; cnm += cache.nm; cth += cache.th; ctm += cache.tm;
		}
#pragma omp critical
		{ gc.tiles += lgc.tiles; gc.node_steps += lgc.node_steps; gc.tri_group_tests += lgc.tri_group_tests; gc.leaves += lgc.leaves; gc.pushes += lgc.pushes; gc.pops += lgc.pops; gc.culled += lgc.culled; gc.t_geom += lgc.t_geom; gc.t_acc += lgc.t_acc; gc.t_edge += lgc.t_edge; gc.t_edge_wrong += lgc.t_edge_wrong; gc.t_far += lgc.t_far; gnh += cnh; gnm += cnm; gth += cth; gtm += ctm; pc.tiles += lpc.tiles; pc.node_rounds += lpc.node_rounds; pc.nodes_tested += lpc.nodes_tested; pc.tri_steps += lpc.tri_steps; pc.leaf_rounds += lpc.leaf_rounds; pc.pushes += lpc.pushes; pc.pops += lpc.pops; pc.culled += lpc.culled; c.tiles += lc.tiles; for (int k = 0; k < 5; k++) c.steps[k] += lc.steps[k]; c.pops += lc.pops; c.pops_culled += lc.pops_culled; c.tri_steps += lc.tri_steps; c.pushes += lc.pushes;
		  c.lane_nodes += lc.lane_nodes; c.entries += lc.entries; c.entry_culled += lc.entry_culled; c.pre_steps += lc.pre_steps; c.blocks += lc.blocks; c.entry_list += lc.entry_list; tsum += ls; mism += lm; }
	}
	if (PK_TILEBEAM == 20) printf("  of the triangle tests per group: %.1f %% meet a ray geometrically, %.1f %% are accepted by a ray, %.1f %% met but behind every hit; the beam's edge test would skip %.1f %% (wrongly: %llu)\n",
		100.0 * gc.t_geom / gc.tri_group_tests, 100.0 * gc.t_acc / gc.tri_group_tests, 100.0 * gc.t_far / gc.tri_group_tests, 100.0 * gc.t_edge / gc.tri_group_tests, (unsigned long long)gc.t_edge_wrong);
	if (PK_TILEBEAM == 20) { const double T3 = (double)gc.tiles; printf("  two ray groups per wave (%d lanes): per tile: node steps %.2f leaves %.2f triangle tests per group %.2f pushes %.2f pops %.2f (culled %.2f) | t mismatches %llu\n", PK_LANES, gc.node_steps / T3, gc.leaves / T3, gc.tri_group_tests / T3, gc.pushes / T3, gc.pops / T3, gc.culled / T3, (unsigned long long)mism); return; }
	if (PK_TILEBEAM >= 2) { const double T2 = (double)pc.tiles; printf("  beam, %d nodes per round: per tile: node rounds %.2f (nodes tested %.2f) leaf rounds %.2f tri steps %.2f pushes %.2f pops %.2f (culled %.2f) | t mismatches %llu\n", PK_TILEBEAM, pc.node_rounds / T2, pc.nodes_tested / T2, pc.leaf_rounds / T2, pc.tri_steps / T2, pc.pushes / T2, pc.pops / T2, pc.culled / T2, (unsigned long long)mism); return; }
	const double T = (double)c.tiles; const uint64_t st = c.steps[0] + c.steps[1] + c.steps[2] + c.steps[3] + c.steps[4];
	printf("  packet %d lanes, entry depth %d, order %d: per tile: node steps %.2f (n_any 0/1/2/3/4: %.2f %.2f %.2f %.2f %.2f) tri steps %.2f pushes %.2f pops %.2f (culled %.2f) lane-nodes/ray %.2f",
		PK_LANES, PK_ENTRY_DEPTH, PK_ORDER, st / T, c.steps[0] / T, c.steps[1] / T, c.steps[2] / T, c.steps[3] / T, c.steps[4] / T, c.tri_steps / T, c.pushes / T, c.pops / T, c.pops_culled / T, (double)c.lane_nodes / (T * PK_LANES));
	if (PK_ENTRY_DEPTH > 0) printf(" | entries taken %.2f (+%.2f cut) list %.1f per block, pre-pass steps %.1f per block", c.entries / T, c.entry_culled / T, (double)c.entry_list / c.blocks, (double)c.pre_steps / c.blocks);
	printf(" | t mismatches %llu\n", (unsigned long long)mism);
	if (PK_CACHE_N || PK_CACHE_T) printf("  software cache per block, tiles in %s order: nodes %d slots hit %.3f | triangles %d slots hit %.3f\n", PK_ZORDER ? "Z" : "row", PK_CACHE_N, (double)gnh / std::max<uint64_t>(1, gnh + gnm), PK_CACHE_T, (double)gth / std::max<uint64_t>(1, gth + gtm));
}

static std::vector<Ray> load_rays(const char *f);
// packets of 64 CONSECUTIVE rays of a file (a sorted batch): wave-level steps per packet, and the same rays one by one
static void trace_any(const Ray &r, Cnt &c) {
	float rd[3] = { 1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2] };
	struct E { float t; int ref; } stack[256]; int sp = 0; int top = 0;
	for (;;) {
		if (top >= 0) {
			const W &w = wide[top]; c.nodes++; float key[8]; int ref[8]; int nh = 0;
			for (int k = 0; k < w.n; k++) { float tn; if (slab(r, rd, w.b[k], r.tmax, tn)) { key[nh] = tn; ref[nh] = w.ref[k]; nh++; } }
			if (ORDERMODE == 0) { for (int i = 1; i < nh; i++) for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); } }
			else if (ORDERMODE == 1 && nh > 1) { int m = 0; for (int i = 1; i < nh; i++) if (key[i] < key[m]) m = i; std::swap(key[0], key[m]); std::swap(ref[0], ref[m]); }   // nearest first, rest in slot order
			// ORDERMODE 2: slot order
			for (int i = nh - 1; i >= 1; i--) { stack[sp].t = key[i]; stack[sp].ref = ref[i]; sp++; }
			if (nh) { top = ref[0]; continue; }
		} else {
			const Leaf &l = leaves[~top]; c.leaves++;
			for (uint32_t p : l.prims) { c.tris++; double t; if (tri_hit(r, &tris[9 * (size_t)p], t) && t > r.tmin && t < r.tmax) { c.hits++; return; } }
		}
		if (!sp) return;
		top = stack[--sp].ref;
	}
}
static void packet_file_lab(const char *file) {
	auto rays = load_rays(file);
	PkCnt c; Cnt sc; uint64_t mixed = 0;
#pragma omp parallel
	{ PkCnt lc; Cnt ls; uint64_t lm = 0;
#pragma omp for schedule(dynamic, 4)
		for (long g = 0; g < (long)(rays.size() / 64); g++) {
			std::vector<Ray> rs(rays.begin() + g * 64, rays.begin() + g * 64 + 64);
			bool uni = true; for (auto &r : rs) for (int a = 0; a < 3; a++) uni &= (r.d[a] < 0) == (rs[0].d[a] < 0);
			lm += !uni;
			trace_packet(rs, lc, nullptr, nullptr);
			for (auto &r : rs) { if (PK_ANY) trace_any(r, ls); else trace(r, ls); }
		}
#pragma omp critical
		{ c.tiles += lc.tiles; for (int k = 0; k < 5; k++) c.steps[k] += lc.steps[k]; c.pops += lc.pops; c.pops_culled += lc.pops_culled; c.tri_steps += lc.tri_steps; c.pushes += lc.pushes; c.lane_nodes += lc.lane_nodes;
		  sc.nodes += ls.nodes; sc.tris += ls.tris; sc.hits += ls.hits; mixed += lm; }
	}
	const double T = (double)c.tiles; const uint64_t st = c.steps[0] + c.steps[1] + c.steps[2] + c.steps[3] + c.steps[4];
	printf("  %s: %s packets of 64 consecutive rays: per packet: node steps %.1f (n_any 0/1/2/3/4: %.1f %.1f %.1f %.1f %.1f) tri steps %.1f pushes %.1f pops %.1f (culled %.1f) lane use in node steps %.3f | mixed-sign packets %.3f | the same rays alone: nodes %.2f tris %.2f hit %.4f\n",
		file, PK_ANY ? "any-hit" : "closest-hit", st / T, c.steps[0] / T, c.steps[1] / T, c.steps[2] / T, c.steps[3] / T, c.steps[4] / T, c.tri_steps / T, c.pushes / T, c.pops / T, c.pops_culled / T,
		(double)c.lane_nodes / (64.0 * st), mixed / T, sc.nodes / (T * 64), sc.tris / (T * 64), sc.hits / (T * 64));
}
// ---- per-lane kernel simulation: the control flow of rtk_trace_kernel (one ray per lane, persistent waves that refill idle lanes)
// on the CPU, counting wave-level node and triangle steps, so that scheduling policies (when to leave the node loop, postponed
// leaves, re-binning rays across the waves of a workgroup) can be priced before they are written for the GPU.
struct LRay {
	Ray r; float rd[3]; float best; bool hit; bool active;
	int top; enum { NONE = INT32_MIN, RETRY = INT32_MIN + 1 };
	struct E { float t; int ref; } stack[128]; int sp;
	int tri_i;                 // next triangle of the leaf `top`
	int pend[8]; float pendt[8]; int npend;     // postponed leaves (nearest first is not kept: order of arrival)
};
struct LCnt { uint64_t rays = 0, node_steps = 0, tri_steps = 0, pop_trips = 0, lane_nodes = 0, lane_tris = 0, refills = 0, rebins = 0, moved = 0; };
static inline bool l_is_node(int top) { return top >= 0; }
static inline bool l_is_leaf(int top) { return top < 0 && top != LRay::NONE && top != LRay::RETRY; }
static void l_start(LRay &s, const Ray &r) { s.r = r; for (int a = 0; a < 3; a++) s.rd[a] = 1.0f / r.d[a]; s.best = r.tmax; s.hit = false; s.active = true; s.top = 0; s.sp = 0; s.tri_i = 0; s.npend = 0; }
static void l_pop(LRay &s) { if (s.sp == 0) { s.top = LRay::NONE; return; } s.sp--; s.top = s.stack[s.sp].t > s.best ? (int)LRay::RETRY : s.stack[s.sp].ref; s.tri_i = 0; }
static bool L_NODE_DEFER_POP = false;
static void l_node(LRay &s) {
	const W &w = wide[s.top]; float key[8]; int ref[8]; int nh = 0;
	for (int k = 0; k < w.n; k++) {
		float tn = s.r.tmin, tf = s.best;
		for (int a = 0; a < 3; a++) { float t0 = (w.b[k].mn[a] - s.r.o[a]) * s.rd[a], t1 = (w.b[k].mx[a] - s.r.o[a]) * s.rd[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
		if (tn <= tf) { key[nh] = tn; ref[nh] = w.ref[k]; nh++; }
	}
	for (int i = 1; i < nh; i++) for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); }
	for (int i = nh - 1; i >= 1; i--) { s.stack[s.sp].t = key[i]; s.stack[s.sp].ref = ref[i]; s.sp++; }
	if (nh) { s.top = ref[0]; s.tri_i = 0; } else if (L_NODE_DEFER_POP) s.top = LRay::RETRY; else l_pop(s);
}
// one triangle of the leaf `leaf`; returns true when the leaf is finished
static bool l_tri(LRay &s, int leaf, int &i) {
	const Leaf &l = leaves[~leaf];
	double t; const uint32_t p = l.prims[i];
	if (tri_hit(s.r, &tris[9 * (size_t)p], t) && t > s.r.tmin && t < s.best) { s.best = (float)t; s.hit = true; }
	return ++i >= (int)l.prims.size();
}

static int WS_NODE_EXIT = 32, WS_REFILL_MIN = 8, WS_WAVES = 32, WS_GROUP = 4, WS_POSTPONE = 0, WS_REBIN_EVERY = 1, WS_TRI_MIN = 1;

// mode 0: the kernel as it stands
static void wave_lab_current(const std::vector<Ray> &rays, LCnt &c) {
	size_t next = 0;   // chunks of 64 in order, dealt to whichever wave asks next
	struct Wave { LRay l[64]; size_t w_next = 0, w_end = 0; bool done = false; };
	std::vector<Wave> waves(WS_WAVES);
	for (auto &w : waves) for (auto &l : w.l) { l.active = false; l.top = LRay::NONE; }
	size_t live = waves.size();
	while (live) for (auto &w : waves) {
		if (w.done) continue;
		// ---- refill
		int n_idle = 0; for (auto &l : w.l) n_idle += !l.active;
		const bool pool_left = w.w_next < w.w_end || next < rays.size();
		if (n_idle == 64 || (n_idle >= WS_REFILL_MIN && pool_left)) {
			if (w.w_next >= w.w_end && next < rays.size()) { w.w_next = next; w.w_end = std::min(rays.size(), next + 64); next = w.w_end; }
			size_t avail = w.w_end - w.w_next;
			if (avail) { c.refills++; for (auto &l : w.l) if (!l.active && w.w_next < w.w_end) { l_start(l, rays[w.w_next++]); c.rays++; } }
			bool any = false; for (auto &l : w.l) any |= l.active;
			if (!any) { if (!(w.w_next < w.w_end || next < rays.size())) { w.done = true; live--; } continue; }
		}
		// ---- node loop
		for (;;) {
			for (auto &l : w.l) if (l.active && l.top == LRay::RETRY) l_pop(l);
			int n_node = 0, n_leaf = 0, n_retry = 0;
			for (auto &l : w.l) if (l.active) { n_node += l_is_node(l.top); n_leaf += l_is_leaf(l.top); n_retry += l.top == LRay::RETRY; }
			if (n_node == 0) { if (n_retry) { c.pop_trips++; continue; } break; }
			if (n_node < WS_NODE_EXIT && n_leaf) break;
			c.node_steps++;
			for (auto &l : w.l) if (l.active && l_is_node(l.top)) { c.lane_nodes++; l_node(l); }
		}
		// ---- leaves: one triangle per step, the wave leaves when its longest leaf is done
		for (;;) {
			int n = 0;
			for (auto &l : w.l) if (l.active && l_is_leaf(l.top)) { n++; c.lane_tris++; if (l_tri(l, l.top, l.tri_i)) l_pop(l); else continue; }
			if (!n) break;
			c.tri_steps++;
			bool more = false; for (auto &l : w.l) if (l.active && l_is_leaf(l.top) && l.tri_i > 0) more = true;
			if (!more) break;
		}
		for (auto &l : w.l) if (l.active && l.top == LRay::NONE) l.active = false;
	}
}

// mode 1: a workgroup of WS_GROUP waves re-bins its rays by state before every WS_REBIN_EVERY-th step: node rays to the low lanes,
// leaf rays behind them; a wave does the step its majority wants (ties: node)
static void wave_lab_rebin(const std::vector<Ray> &rays, LCnt &c) {
	const int G = WS_GROUP * 64;
	size_t next = 0;
	const int ngroups = std::max(1, WS_WAVES / WS_GROUP);
	std::vector<std::vector<LRay>> groups(ngroups, std::vector<LRay>(G));
	for (auto &g : groups) for (auto &l : g) { l.active = false; l.top = LRay::NONE; }
	std::vector<bool> done(ngroups, false); int live = ngroups; std::vector<uint64_t> round(ngroups, 0);
	while (live) for (int gi = 0; gi < ngroups; gi++) {
		if (done[gi]) continue;
		auto &g = groups[gi];
		// retire + refill (per wave, when at least WS_REFILL_MIN of its lanes are idle)
		for (auto &l : g) if (l.active && l.top == LRay::NONE) l.active = false;
		for (int w = 0; w < WS_GROUP; w++) {
			int n_idle = 0; for (int i = 0; i < 64; i++) n_idle += !g[w * 64 + i].active;
			if (n_idle >= WS_REFILL_MIN && next < rays.size()) { c.refills++; for (int i = 0; i < 64; i++) { LRay &l = g[w * 64 + i]; if (!l.active && next < rays.size()) { l_start(l, rays[next++]); c.rays++; } } }
		}
		bool any = false; for (auto &l : g) any |= l.active;
		if (!any) { if (next >= rays.size()) { done[gi] = true; live--; } continue; }
		for (auto &l : g) if (l.active && l.top == LRay::RETRY) l_pop(l);
		if (WS_REBIN_EVERY > 0 && round[gi]++ % WS_REBIN_EVERY == 0) {
			// stable partition: node | retry/none | leaf  (leaves at the far end, so that both ends fill whole waves)
			std::vector<LRay> a, b, d; a.reserve(G); 
			for (auto &l : g) { if (l.active && l_is_node(l.top)) a.push_back(l); else if (l.active && l_is_leaf(l.top)) d.push_back(l); else b.push_back(l); }
			size_t k = 0; for (auto &l : a) g[k++] = l; for (auto &l : b) g[k++] = l; for (auto &l : d) g[k++] = l;
			c.rebins++;
		}
		for (int w = 0; w < WS_GROUP; w++) {
			int n_node = 0, n_leaf = 0;
			for (int i = 0; i < 64; i++) { LRay &l = g[w * 64 + i]; if (l.active) { n_node += l_is_node(l.top); n_leaf += l_is_leaf(l.top); } }
			if (n_node == 0 && n_leaf == 0) { c.pop_trips++; continue; }
			if (n_node >= n_leaf && n_node >= 1 && !(n_leaf >= WS_TRI_MIN && n_node < WS_NODE_EXIT)) {
				c.node_steps++;
				for (int i = 0; i < 64; i++) { LRay &l = g[w * 64 + i]; if (l.active && l_is_node(l.top)) { c.lane_nodes++; l_node(l); } }
			} else if (n_leaf) {
				c.tri_steps++;
				for (int i = 0; i < 64; i++) { LRay &l = g[w * 64 + i]; if (l.active && l_is_leaf(l.top)) { c.lane_tris++; if (l_tri(l, l.top, l.tri_i)) l_pop(l); } }
			} else {
				c.node_steps++;
				for (int i = 0; i < 64; i++) { LRay &l = g[w * 64 + i]; if (l.active && l_is_node(l.top)) { c.lane_nodes++; l_node(l); } }
			}
		}
	}
}

// mode 2: postponed leaves: a lane that reaches a leaf parks it (up to WS_POSTPONE of them) and pops on; parked leaves are
// tested when at least WS_TRI_MIN lanes are blocked (list full or stack empty) or fewer than WS_NODE_EXIT lanes want nodes
static void wave_lab_postpone(const std::vector<Ray> &rays, LCnt &c) {
	size_t next = 0;
	struct Wave { LRay l[64]; bool done = false; };
	std::vector<Wave> waves(WS_WAVES);
	for (auto &w : waves) for (auto &l : w.l) { l.active = false; l.top = LRay::NONE; l.npend = 0; }
	size_t live = waves.size();
	while (live) for (auto &w : waves) {
		if (w.done) continue;
		int n_idle = 0; for (auto &l : w.l) n_idle += !l.active;
		if ((n_idle == 64 || n_idle >= WS_REFILL_MIN) && next < rays.size()) { c.refills++; for (auto &l : w.l) if (!l.active && next < rays.size()) { l_start(l, rays[next++]); c.rays++; } }
		bool any = false; for (auto &l : w.l) any |= l.active;
		if (!any) { if (next >= rays.size()) { w.done = true; live--; } continue; }
		// park leaves / pops
		for (auto &l : w.l) if (l.active) {
			if (l.top == LRay::RETRY) l_pop(l);
			if (l_is_leaf(l.top) && l.npend < WS_POSTPONE) { l.pend[l.npend++] = l.top; l_pop(l); }
		}
		int n_node = 0, n_blocked = 0, n_pend = 0;
		for (auto &l : w.l) if (l.active) { n_node += l_is_node(l.top); const bool blocked = l_is_leaf(l.top) || (l.top == LRay::NONE && l.npend); n_blocked += blocked; n_pend += l.npend > 0 || l_is_leaf(l.top); }
		if (n_node && !(n_blocked >= WS_TRI_MIN && n_node < WS_NODE_EXIT) ) {
			c.node_steps++;
			for (auto &l : w.l) if (l.active && l_is_node(l.top)) { c.lane_nodes++; l_node(l); }
		} else if (n_pend) {
			// every lane with a parked (or current) leaf tests ONE leaf's triangles; steps = the longest
			int steps = 0;
			for (auto &l : w.l) if (l.active) {
				int leaf; bool cur = false;
				if (l.npend) { leaf = l.pend[0]; for (int k = 1; k < l.npend; k++) l.pend[k - 1] = l.pend[k]; l.npend--; }
				else if (l_is_leaf(l.top)) { leaf = l.top; cur = true; }
				else continue;
				int i = 0, n = 0; for (;;) { n++; c.lane_tris++; if (l_tri(l, leaf, i)) break; }
				steps = std::max(steps, n);
				if (cur) l_pop(l);
				else if (l.top == LRay::NONE && l.sp == 0 && l.npend == 0) {}
				// a node or entry that the new best culls is dropped by the next pop / slab test as usual
				if (l_is_node(l.top) == false && l.top != LRay::NONE && l.top != LRay::RETRY && false) {}
			}
			c.tri_steps += steps;
		} else c.pop_trips++;
		for (auto &l : w.l) if (l.active && l.top == LRay::NONE && l.npend == 0) l.active = false;
	}
}

// mode 3: the ray pool. Ray state lives in LDS slots; every trip a wave claims up to 64 rays that all want the same kind of step
// (node / triangle / set-up of a new ray) from the workgroup's queues, does that step at (nearly) full lane use, and hands every ray
// to the queue of its next state. Discrete-event simulation: WS_WAVES waves of one workgroup on four SIMDs, a step = compute
// (vector instructions x 4 clk, exclusive on its SIMD) after a memory wait of WS_LAT clk during which the SIMD is free.
static int WS_POPS = 3;
static int WS_POOL = 1280, WS_LAT = 1500, WS_LEAF_FIRST = 1, WS_MIN_BATCH = 64;
static void wave_lab_pool(const std::vector<Ray> &rays, LCnt &c, double &simd_busy, double &total_clk, double &valu) {
	const int P = WS_POOL, NW = WS_WAVES; L_NODE_DEFER_POP = !(WS_POPS & 2);
	std::vector<LRay> pool(P);
	std::vector<int> q[3];   // 0 node, 1 leaf, 2 free
	for (int i = 0; i < P; i++) { pool[i].active = false; q[2].push_back(i); }
	size_t next = 0; size_t retired = 0;
	struct Wv { double t; int phase; int kind; std::vector<int> batch; };   // phase 0: idle (wants a batch at time t); 1: waiting for memory until t, then needs its SIMD
	std::vector<Wv> wv(NW); for (auto &w : wv) { w.t = 0; w.phase = 0; }
	double simd_free[4] = { 0, 0, 0, 0 }, busy = 0; valu = 0;
	const double COST_ISSUE = 45, COST_NODE = 135, COST_TRI = 115, COST_SETUP = 200, COST_POP = 12;
	double now = 0; uint64_t guard = 0;
	while (retired < rays.size() && guard++ < (1ull << 40)) {
		// next event: the wave with the smallest time
		int wi = 0; for (int i = 1; i < NW; i++) if (wv[i].t < wv[wi].t) wi = i;
		Wv &w = wv[wi]; now = w.t; const int sd = wi & 3;
		if (w.phase == 0) {
			// claim a batch
			const size_t nn = q[0].size(), nl = q[1].size(), nf = next < rays.size() ? q[2].size() : 0;
			int kind = -1;
			const size_t mb = (size_t)WS_MIN_BATCH;
			if (WS_LEAF_FIRST && nl >= mb) kind = 1; else if (nn >= mb) kind = 0; else if (nl >= mb) kind = 1; else if (nf >= mb) kind = 2;
			else {
				// nothing fills a wave: take the fullest queue only if nobody else is about to deliver (all other waves idle) -- else wait
				bool others_busy = false; for (int i = 0; i < NW; i++) if (i != wi && wv[i].phase == 1) others_busy = true;
				const size_t best = std::max(nn, std::max(nl, nf));
				if (best == 0 || (others_busy && best < mb)) { w.t = now + 200; if (!others_busy && best == 0) w.t = now + 1000; continue; }
				kind = nn == best ? 0 : (nl == best ? 1 : 2);
			}
			auto &Q = q[kind]; const size_t k = std::min<size_t>(64, Q.size());
			w.batch.assign(Q.begin(), Q.begin() + k); Q.erase(Q.begin(), Q.begin() + k);
			w.kind = kind;
			// issue part on the SIMD, then the memory wait
			const double st = std::max(now, simd_free[sd]); simd_free[sd] = st + COST_ISSUE * 4; busy += COST_ISSUE * 4; valu += COST_ISSUE;
			w.t = simd_free[sd] + (kind == 2 ? WS_LAT : WS_LAT); w.phase = 1;
		} else {
			// compute part
			double cost = w.kind == 0 ? COST_NODE : (w.kind == 1 ? COST_TRI : COST_SETUP);
			int extra_pops = 0;
			if (w.kind == 0) { c.node_steps++; cost += COST_POP; for (int s : w.batch) { LRay &l = pool[s]; if (l.top == LRay::RETRY) l_pop(l); if (l_is_node(l.top)) { c.lane_nodes++; l_node(l); } } }
			else if (w.kind == 1) { c.tri_steps++; for (int s : w.batch) { LRay &l = pool[s]; c.lane_tris++; if (l_tri(l, l.top, l.tri_i)) { if (WS_POPS & 1) l_pop(l); else l.top = LRay::RETRY; } } }
			else { c.refills++; for (int s : w.batch) { if (next < rays.size()) { l_start(pool[s], rays[next++]); c.rays++; } else pool[s].top = LRay::NONE, pool[s].active = false; } }
			cost += COST_POP * extra_pops;
			const double st = std::max(now, simd_free[sd]); simd_free[sd] = st + cost * 4; busy += cost * 4; valu += cost;
			for (int s : w.batch) { LRay &l = pool[s];
				if (l_is_node(l.top) || l.top == LRay::RETRY) q[0].push_back(s); else if (l_is_leaf(l.top)) q[1].push_back(s);
				else { if (l.active) { retired++; l.active = false; } q[2].push_back(s); } }
			w.t = simd_free[sd]; w.phase = 0;
		}
	}
	total_clk = now; simd_busy = busy / (4.0 * now);
}

// mode 4: every wave keeps WS_PER_LANE rays per lane in LDS (slot = k * 64 + lane: no ray ever changes lane, so no queues, no
// atomics, no waiting on other waves, conflict-free LDS); each trip the wave picks the kind of step most lanes can take part in and
// every lane works on the first of its rays that wants that step.
static int WS_PER_LANE = 2, WS_LEAF_MIN = 40, WS_SETUP_MIN = 24;
static void wave_lab_private(const std::vector<Ray> &rays, LCnt &c, double &valu) {
	const int K = WS_PER_LANE;
	size_t next = 0; valu = 0;
	struct Wave { std::vector<LRay> l; bool done = false; };
	std::vector<Wave> waves(WS_WAVES);
	for (auto &w : waves) { w.l.resize(64 * K); for (auto &l : w.l) { l.active = false; l.top = LRay::NONE; } }
	size_t live = waves.size();
	const double COST_ISSUE = 40, COST_NODE = 145, COST_TRI = 115, COST_SETUP = 200;
	while (live) for (auto &w : waves) {
		if (w.done) continue;
		int cN = 0, cL = 0, cF = 0, any = 0;
		for (int i = 0; i < 64; i++) { bool n = false, lf = false, f = false; for (int k = 0; k < K; k++) { LRay &l = w.l[k * 64 + i]; if (!l.active) f = true; else { any++; if (l_is_node(l.top) || l.top == LRay::RETRY) n = true; else if (l_is_leaf(l.top)) lf = true; } } cN += n; cL += lf; cF += f; }
		const bool more = next < rays.size();
		if (!any && !more) { w.done = true; live--; continue; }
		int kind;
		if (cL >= WS_LEAF_MIN) kind = 1; else if (more && cF >= WS_SETUP_MIN && (cF >= 48 || cN < 48)) kind = 2; else if (cN >= 1 && cN >= cL) kind = 0; else if (cL) kind = 1; else if (more && cF) kind = 2; else kind = 0;
		valu += COST_ISSUE;
		if (kind == 0) { c.node_steps++; valu += COST_NODE; for (int i = 0; i < 64; i++) for (int k = 0; k < K; k++) { LRay &l = w.l[k * 64 + i]; if (l.active && (l_is_node(l.top) || l.top == LRay::RETRY)) { if (l.top == LRay::RETRY) l_pop(l); if (l_is_node(l.top)) { c.lane_nodes++; l_node(l); } break; } } }
		else if (kind == 1) { c.tri_steps++; valu += COST_TRI; for (int i = 0; i < 64; i++) for (int k = 0; k < K; k++) { LRay &l = w.l[k * 64 + i]; if (l.active && l_is_leaf(l.top)) { c.lane_tris++; if (l_tri(l, l.top, l.tri_i)) l_pop(l); break; } } }
		else { c.refills++; valu += COST_SETUP; for (int i = 0; i < 64; i++) for (int k = 0; k < K; k++) { LRay &l = w.l[k * 64 + i]; if (!l.active) { if (next < rays.size()) { l_start(l, rays[next++]); c.rays++; } break; } } }
		for (auto &l : w.l) if (l.active && l.top == LRay::NONE) l.active = false;
	}
}

static void wave_lab(int mode) {
	auto rays = load_rays("rays_inc.bin");
	LCnt c;
	if (mode == 3) {
		double sb, clk, valu; wave_lab_pool(rays, c, sb, clk, valu);
		const double per64 = (double)c.rays / 64.0;
		printf("  ray pool (%d slots, %d waves, latency %d clk, leaf first %d, min batch %d): per 64 rays: node trips %.2f tri trips %.2f set-up trips %.2f | lane use: node %.3f tri %.3f | vector instructions per 64 rays %.0f | SIMD busy %.3f | clk per 64 rays per CU %.0f\n",
			WS_POOL, WS_WAVES, WS_LAT, WS_LEAF_FIRST, WS_MIN_BATCH, c.node_steps / per64, c.tri_steps / per64, c.refills / per64, c.lane_nodes / (64.0 * c.node_steps), c.lane_tris / (64.0 * c.tri_steps), valu / per64, sb, clk / per64);
		return;
	}
	if (mode == 4) {
		double valu; wave_lab_private(rays, c, valu);
		const double per64 = (double)c.rays / 64.0;
		printf("  %d rays per lane (leaf trip at %d lanes, set-up at %d): per 64 rays: node trips %.2f tri trips %.2f set-up trips %.2f | lane use: node %.3f tri %.3f | vector instructions per 64 rays %.0f\n",
			WS_PER_LANE, WS_LEAF_MIN, WS_SETUP_MIN, c.node_steps / per64, c.tri_steps / per64, c.refills / per64, c.lane_nodes / (64.0 * c.node_steps), c.lane_tris / (64.0 * c.tri_steps), valu / per64);
		return;
	}
	if (mode == 0) wave_lab_current(rays, c); else if (mode == 1) wave_lab_rebin(rays, c); else wave_lab_postpone(rays, c);
	const double per64 = (double)c.rays / 64.0;
	printf("  wave lab mode %d (node_exit %d refill_min %d waves %d group %d postpone %d rebin_every %d tri_min %d): per 64 rays: node steps %.2f tri steps %.2f pop trips %.2f refills %.2f | per ray: nodes %.2f tris %.2f | lane use: node %.3f tri %.3f | cost(190/130) %.0f\n",
		mode, WS_NODE_EXIT, WS_REFILL_MIN, WS_WAVES, WS_GROUP, WS_POSTPONE, WS_REBIN_EVERY, WS_TRI_MIN, c.node_steps / per64, c.tri_steps / per64, c.pop_trips / per64, c.refills / per64,
		(double)c.lane_nodes / c.rays, (double)c.lane_tris / c.rays, c.lane_nodes / (64.0 * c.node_steps), c.lane_tris / (64.0 * c.tri_steps),
		(190.0 * c.node_steps + 130.0 * c.tri_steps + 20.0 * c.pop_trips + 150.0 * c.refills) / per64);
}

static std::vector<Ray> load_rays(const char *f) { FILE *fp = fopen(f, "rb"); fseek(fp, 0, SEEK_END); long s = ftell(fp); fseek(fp, 0, SEEK_SET); std::vector<Ray> r(s / 32); if (fread(r.data(), 32, r.size(), fp) != r.size()) abort(); fclose(fp); return r; }

static double tree_sah() {
	// SAH of the BVH4: sum over wide nodes of child areas (node test cost per child visited ~ area of the wide node's own box) -- report classic binary SAH instead
	double ra = harea(bin[root].b), s = 0;
	for (auto &w : wide) for (int k = 0; k < w.n; k++) s += harea(w.b[k]) / ra * (w.ref[k] >= 0 ? 1.0 : (double)leaves[~w.ref[k]].prims.size());
	return s;
}

int main(int argc, char **argv) {
	const char *trisf = "tris_1m.f32"; std::string builder = "lbvh"; int R = 16; int ws_mode = -1; std::vector<const char *> pk_files; bool skip_rays = false;
	for (int i = 1; i < argc; i++) {
		if (!strcmp(argv[i], "-b")) builder = argv[++i];
		else if (!strcmp(argv[i], "-r")) R = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-cn")) CN = atof(argv[++i]);
		else if (!strcmp(argv[i], "-ml")) MAXLEAF = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-cm")) COLLAPSE_MODE = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-bob")) BIN_OVER_BOX = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-cc")) COLLAPSE_CRIT = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-t")) trisf = argv[++i];
		else if (!strcmp(argv[i], "-T")) HYB_T = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-ks")) KEYSHIFT = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-w")) WIDTH = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-om")) ORDERMODE = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-bins")) SAH_BINS = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-ft")) FORCE_TILE = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-pk")) PK_LANES = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-pe")) PK_ENTRY_DEPTH = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-po")) PK_ORDER = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-pb")) PK_BLOCKS = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-ws")) ws_mode = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-pf")) pk_files.push_back(argv[++i]);
		else if (!strcmp(argv[i], "-pa")) PK_ANY = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-pm")) PK_BEAM = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-tb")) PK_TILEBEAM = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-cn_")) PK_CACHE_N = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-ct_")) PK_CACHE_T = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-pz")) PK_ZORDER = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-norays")) skip_rays = true;
		else if (!strcmp(argv[i], "-wne")) WS_NODE_EXIT = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wrm")) WS_REFILL_MIN = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-ww")) WS_WAVES = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wg")) WS_GROUP = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wp")) WS_POSTPONE = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wre")) WS_REBIN_EVERY = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wtm")) WS_TRI_MIN = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wpool")) WS_POOL = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wpops")) WS_POPS = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wpl")) WS_PER_LANE = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wlm")) WS_LEAF_MIN = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wsm")) WS_SETUP_MIN = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wlat")) WS_LAT = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wlf")) WS_LEAF_FIRST = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-wmb")) WS_MIN_BATCH = atoi(argv[++i]);
	}
	{ FILE *fp = fopen(trisf, "rb"); fseek(fp, 0, SEEK_END); long s = ftell(fp); fseek(fp, 0, SEEK_SET); tris.resize(s / 4); if (fread(tris.data(), 4, tris.size(), fp) != tris.size()) abort(); fclose(fp); N = tris.size() / 9; }
	morton_sort();
	auto t0 = std::chrono::steady_clock::now();
	if (builder == "lbvh") { bin.clear(); bin.reserve(N); root = lbvh_rec(0, (int)N - 1); refit_rec(root); }
	else if (builder == "sah") { bin.clear(); bin.reserve(N); std::vector<int> idx(N); for (size_t i = 0; i < N; i++) idx[i] = (int)i; root = sah_build(idx, 0, (int)N); refit_rec(root); }
	else if (builder == "hyb") { bin.clear(); bin.reserve(N); root = hyb_rec(0, (int)N - 1); refit_rec(root); }
	else if (builder == "mswp") { bin.clear(); bin.reserve(N); root = mswp_rec(0, (int)N - 1); refit_rec(root); }
	else if (builder == "grp") { bin.clear(); bin.reserve(N); root = grp_rec(0, (int)((N + HYB_T - 1) / HYB_T) - 1); refit_rec(root); }
	else build_ploc(R);
	double bt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	sah_rec(root);
	collapse();
	size_t ntl = 0; for (auto &l : leaves) ntl += l.prims.size();
	printf("%s R=%d cn=%.2f ml=%u cm=%d: build %.2fs wide %zu leaves %zu (%.2f tris/leaf) sah4 %.2f\n", builder.c_str(), R, CN, MAXLEAF, COLLAPSE_MODE, bt, wide.size(), leaves.size(), (double)ntl / leaves.size(), tree_sah());
	if (!skip_rays) for (const char *rf : { "rays_coh.bin", "rays_inc.bin" }) {
		auto rays = load_rays(rf); Cnt c;
#pragma omp parallel
		{ Cnt lc;
#pragma omp for schedule(dynamic, 256)
			for (long i = 0; i < (long)rays.size(); i++) trace(rays[i], lc);
#pragma omp critical
			{ c.nodes += lc.nodes; c.leaves += lc.leaves; c.tris += lc.tris; c.hits += lc.hits; for (int k = 0; k < 40; k++) { c.sp_hist[k] += lc.sp_hist[k]; c.max_hist[k] += lc.max_hist[k]; } } }
		double n = (double)rays.size();
		printf("  %-14s nodes %.2f leaves %.2f tris %.2f hit %.4f\n", rf, c.nodes / n, c.leaves / n, c.tris / n, c.hits / n);
		printf("     stack depth after a node step (share of steps), depth 0..: "); for (int k = 0; k < 24; k++) printf("%.3f ", (double)c.sp_hist[k] / c.nodes); printf("\n     deepest stack of a ray (share of rays): "); for (int k = 0; k < 24; k++) printf("%.3f ", c.max_hist[k] / n); printf("\n");
	}
	if (PK_BLOCKS) packet_lab(4096, 4096, PK_BLOCKS);
	if (ws_mode >= 0) wave_lab(ws_mode);
	if (!pk_files.empty()) { wperm.assign(wide.size() * 8, 0xe4); for (const char *f : pk_files) packet_file_lab(f); }
	return 0;
}
