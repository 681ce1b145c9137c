// multifrontal.hip -- nested-dissection (multifrontal) elimination of the periodic micro problem for LARGE plane blocks (3D elasticity from
// 6^3 micro cells -- BASELINE configurations C4 / C5: 16^3 micro cells x 3 components = 12,288 unknowns per macro cell -- and large 2D
// meshes; the thresholds are set in blocked_workspace_create from measurements).
//
// Why: the block-cyclic plane elimination of blocked.hip carries a dense b x b arrow through n - 1 steps, (6 (n-1) + 2) b^3 model flops
// (41.7 GFLOP per C4 / C5 cell); a sparse Cholesky under nested dissection needs 11.3 GFLOP (profiles/fref.json).  Here the torus is
// dissected geometrically -- two planes per periodic direction, one per open direction, down to 3 x 3 x 3-node leaves -- and every
// supernode (separator or leaf) is eliminated as a dense FRONT
//
//        F = [ F11  F12 ]   s = the supernode's unknowns, r = the later-eliminated unknowns it touches (its "boundary")
//            [ F21  F22 ]
//        N = F11^-1 ;  F12 <- N F21^T ;  F22 <- F22 - F21 F12 (lower tiles)
//
// with the SAME batched fp64-MFMA building blocks as the plane elimination (blocked.hip: k_gemm_tile, recursive block inverse).  All macro
// cells share the structure, and fronts of equal shape at equal height of the elimination tree are independent, so every step is ONE
// batched launch over (cells x fronts of that shape).
//   * The t canonical load vectors ride as a BORDER: 8 extra rows at the end of every front's boundary block (row m = load case m), so
//     the same two GEMMs that update the boundary matrix also update the load rows and accumulate  -R_s N R_s^T  in the 8 x 8 corner; the
//     corner travels up the tree with the update matrices and arrives at the root as  -B^T K^+ B:  A_H = C0 + corner
//     (= C0 - B^T K^+ B; reference forms hmm.py:644-667 / 759-789 / 887-922 / 1024-1067).
//   * Extend-add without a pass of its own: F22 of a front holds nothing but its children's update matrices (original stencil entries
//     only ever sit in the columns a front eliminates), so it is never built -- the Schur-update GEMM reads "C" THROUGH the child -> parent
//     index maps (GatherC epilogue of k_gemm_tile) and writes the front's update matrix once.  Only F11 / F21 (the eliminated columns)
//     are written by k_mf_build: stencil entries + the children's entries, every entry exactly once -- no memset, no atomics.
//   * Nodes are numbered in elimination order everywhere, so child -> parent maps are monotone and only lower triangles are ever needed.
//   * Every chunk of cells runs as two to four pieces side by side on as many streams (the caller's and plan-owned ones, mf_solve).
// Gauge: the last node is pinned in the root front (cell_problem.py:349-361).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <vector>

#include "../../include/hommx_hip.h"
#include "blocked_internal.h"
#include "kernels.h"
#include "mf_front.h"

namespace hommx {

#define MTRY(expr)                                                                       \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess) {                                                             \
      g_berr = std::string(#expr) + ": " + hipGetErrorString(e__);                       \
      return e__ == hipErrorOutOfMemory ? HOMMX_ENOMEM : HOMMX_EHIP;                     \
    }                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------
// plan: symbolic analysis (host) + index tables (device)
// ---------------------------------------------------------------------------------------------------------------
struct MfGroup {
  int height = 0, ns = 0, nr = 0;   // nodes eliminated here / boundary nodes
  int sp = 0, rb = 0, rp = 0, L = 0;   // padded eliminated unknowns (multiple of 32); boundary unknowns nr * bs; rb + border, padded to 16; sp + rp
  int nf = 0;                       // fronts of this shape at this height
  long long offF = 0;               // per-cell arena offset (doubles): F [nf][L][L]
  int pinpos = -1;                  // root only: local node whose unknowns are pinned (gauge)
  // device tables, [nf] x ...
  int32_t* d_nodes = nullptr;       // [nf][nloc]      global (periodic) node of local node
  int8_t* d_code = nullptr;         // [nf][nloc][ns]  stencil code of (row node i, column node j < ns), -1: no coupling
  int32_t* d_cpos = nullptr;        // [nf][2][nloc]   position of local node in child c's boundary list, -1: not there
  int32_t* d_dpos = nullptr;        // [nf][2][rp]     unknown of child c's boundary block for boundary unknown p of this front, -1: none
  MfChild* d_child = nullptr;       // [nf][2]
  // register-resident route of the group (mf_front.h: k_mf_front), when the whole front fits the registers of one workgroup
  bool front = false;
  int s16 = 0, T = 0, P = 0, ntiles = 0;   // 16-granular padded s, tiles per dimension, panels, upper tiles
  int R0 = 0, nreg = 0;             // tile rows [0, R0) live in LDS (fronts of more than MFF_REG_TILES upper tiles), nreg register tiles (rows >= R0)
  int32_t* d_upos = nullptr;        // [nf][2][16 T]  unknown of the front -> unknown of the child's update matrix, -1: none
  uint16_t* d_tilemap = nullptr;    // [ntiles]       a << 8 | b
  bool has_children = false;
};

struct MfPlan {
  Geo G;
  std::vector<MfGroup> groups;      // in processing order (height ascending)
  int nfronts = 0;
  long long arena_per_cell = 0;     // doubles: fronts, lifetimes overlapped
  long long scratch_per_cell = 0;   // doubles: inverse scratch of the largest group
  double flops_per_cell = 0.0;      // executed dense flops by the model of mf_solve (staged elimination + Schur update) on the padded sizes
  int stage = 192;                  // HOMMX_MF_STAGE: unknowns per elimination stage inside a front (0: one stage)
  int front_max_t = MFF_MAX_T;      // HOMMX_MF_FRONT: most 16-tiles per dimension of a front that takes k_mf_front (0: never; at most MFF_MAX_T)
  bool keep = false;                // corrector plan: every front keeps its own place in the arena (the back substitution reads N and X of all of them)
  long long vbuf_per_cell = 0;      // doubles: solution vectors of the largest group, [nf][MF_BORDER][L] (corrector plan)
  // chunk buffers
  long long chunk = 0;
  double *arena = nullptr, *scratch = nullptr, *vbuf = nullptr;
  double *Kst = nullptr, *Brhs = nullptr, *C0 = nullptr;  // K1 output of this route's chunks (the plane elimination keeps its own)
  // side streams of mf_solve: a chunk runs as up to four pieces side by side (HOMMX_MF_STREAMS = 1: off, 2 .. 4)
  int streams = 4;
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
};

constexpr int MF_BORDER = 8;  // load rows per front (t <= 6)
constexpr int MF_BUILD_ROWS = 16;  // rows of a front one workgroup of k_mf_build writes

// stages of the elimination inside one front (mf_solve): about `stage` unknowns each, multiples of 32
static inline int mf_stages(int sp, int stage) { return (stage <= 0 || sp < 2 * stage - 64) ? 1 : (sp + stage - 1) / stage; }
static inline int mf_stage_size(int sp, int nst, int i) {
  const int base = (sp / 32) / nst, extra = (sp / 32) % nst;  // 32-blocks per stage; the first `extra` stages take one more
  return 32 * (base + (i < extra ? 1 : 0));
}

namespace {

struct SN {
  std::vector<int> nodes, children, bnd;
  int height = 0, group = -1, fidx = -1;
};

struct TreeBuilder {
  int dim, n, leaf_max;
  int split_depth = 0;  // ring cuts at recursion depth <= split_depth are eliminated plane after plane (a chain of two fronts)
  std::vector<SN> sn;
  int coord(int node, int ax) const {
    for (int k = 0; k < ax; ++k) node /= n;
    return node % n;
  }
  // nested dissection of the box [lo, hi) (periodic[ax]: the box is the whole ring in that direction); returns the top supernode
  int rec(const std::vector<int>& sel, const int* lo, const int* hi, const bool* periodic, int depth = 0) {
    if (sel.empty()) return -1;
    int size[3] = {1, 1, 1}, mx = 0;
    for (int a = 0; a < dim; ++a) {
      size[a] = hi[a] - lo[a];
      mx = std::max(mx, size[a]);
    }
    if ((int)sel.size() <= leaf_max || mx <= 2) {
      SN s;
      s.nodes = sel;
      sn.push_back(s);
      return (int)sn.size() - 1;
    }
    int ax = 0;
    for (int a = 1; a < dim; ++a)
      if (size[a] > size[ax]) ax = a;
    int l1[3], h1[3], l2[3], h2[3];
    bool per2[3];
    for (int a = 0; a < 3; ++a) {
      l1[a] = l2[a] = lo[a];
      h1[a] = h2[a] = hi[a];
      per2[a] = periodic[a];
    }
    std::vector<int> a_, b_, sepA, sepB;
    if (periodic[ax]) {  // a ring needs two cuts: the last plane and the middle one
      const int cutA = hi[ax] - 1, cutB = lo[ax] + (size[ax] - 1) / 2;
      for (int v : sel) {
        const int c = coord(v, ax);
        if (c == cutA) sepA.push_back(v);
        else if (c == cutB) sepB.push_back(v);
        else if (c < cutB) a_.push_back(v);
        else b_.push_back(v);
      }
      per2[ax] = false;
      h1[ax] = cutB;
      l2[ax] = cutB + 1;
      h2[ax] = cutA;
    } else {
      const int mid = lo[ax] + size[ax] / 2;
      for (int v : sel) {
        const int c = coord(v, ax);
        if (c == mid) sepA.push_back(v);
        else if (c < mid) a_.push_back(v);
        else b_.push_back(v);
      }
      h1[ax] = mid;
      l2[ax] = mid + 1;
    }
    const int ca = rec(a_, l1, h1, per2, depth + 1), cb = rec(b_, l2, h2, per2, depth + 1);
    // The two planes of a ring cut can be eliminated one after the other (a chain of two fronts: an s-sized inverse costs s^3, two
    // halves a quarter of it) or together.  The chain saves flops but costs one more pass over the r x r boundary matrix, and the
    // update of a front is a rank-s update of that matrix -- memory-bound for small s: only the root (r = 0 / r = s) is split.
    int top = -1;
    if (!sepB.empty() && depth > split_depth) {
      sepA.insert(sepA.end(), sepB.begin(), sepB.end());
      sepB.clear();
    }
    if (!sepB.empty()) {
      SN s;
      s.nodes = sepB;
      if (ca >= 0) s.children.push_back(ca);
      if (cb >= 0) s.children.push_back(cb);
      sn.push_back(s);
      top = (int)sn.size() - 1;
    }
    SN s;
    s.nodes = sepA;
    if (top >= 0) s.children.push_back(top);
    else {
      if (ca >= 0) s.children.push_back(ca);
      if (cb >= 0) s.children.push_back(cb);
    }
    sn.push_back(s);
    return (int)sn.size() - 1;
  }
};

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
  *dst = nullptr;
  if (src.empty()) return 0;
  MTRY(hipMalloc(dst, sizeof(T) * src.size()));
  MTRY(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
  return 0;
}

}  // namespace

double mf_flops_per_cell(const MfPlan* p) { return p ? p->flops_per_cell : 0.0; }

std::string mf_describe(const BlockedWorkspace* ws, const MfPlan* p) {
  if (!p) return "";
  bool t64 = false, t128 = false;
  for (const MfGroup& mg : p->groups) (gemm_tile_size(ws, mg.rp, mg.rp, mg.sp, true) == 128 ? t128 : t64) = true;
  const MfGroup& top = p->groups.back();
  const MfGroup& leaf = p->groups.front();
  int nfront = 0, ffront = 0;
  for (const MfGroup& mg : p->groups)
    if (mg.front) {
      ++nfront;
      ffront += mg.nf;
    }
  char buf[900];
  snprintf(buf, sizeof(buf),
           "multifrontal: nested dissection, %d fronts in %d groups (leaves s = %d, r = %d x %d; root s = %d), stages of %d unknowns, up to %d stream(s) per "
           "chunk; batched launches per group: k_mf_build, recursive block inverse (k_leaf_inverse / k_leaf_inverse_blk<64> + k_gemm_tile), X = N E^T and "
           "column updates on k_gemm_tile<%s>, gathering Schur update k_gemm_tile<%s%s, GATHER> (the dominant kernel)",
           p->nfronts, (int)p->groups.size(), leaf.ns * ws->G.bs, leaf.rb, leaf.nf, top.ns * ws->G.bs, p->stage, std::max(1, std::min(4, p->streams)),
           ws->gemm128_min < (1 << 29) ? "64x64 / 128x128" : "64x64, 4 waves", t64 ? "64x64, 4 waves" : "", t128 ? (t64 ? " / 128x128, 8 waves" : "128x128, 8 waves") : "");
  std::string out = buf;
  if (nfront) {
    snprintf(buf, sizeof(buf), "; %d of the groups (%d of the fronts, at most %d x %d unknowns) in ONE launch each on the register-resident front kernel k_mf_front",
             nfront, ffront, 16 * p->front_max_t, 16 * p->front_max_t);
    out += buf;
  }
  return out;
}

void mf_plan_destroy(MfPlan* p) {
  if (!p) return;
  for (MfGroup& g : p->groups) {
    if (g.d_nodes) (void)hipFree(g.d_nodes);
    if (g.d_code) (void)hipFree(g.d_code);
    if (g.d_cpos) (void)hipFree(g.d_cpos);
    if (g.d_dpos) (void)hipFree(g.d_dpos);
    if (g.d_child) (void)hipFree(g.d_child);
    if (g.d_upos) (void)hipFree(g.d_upos);
    if (g.d_tilemap) (void)hipFree(g.d_tilemap);
  }
  for (double* q : {p->arena, p->scratch, p->vbuf, p->Kst, p->Brhs, p->C0})
    if (q) (void)hipFree(q);
  for (int k = 0; k < 3; ++k) {
    if (p->side[k]) (void)hipStreamDestroy(p->side[k]);
    if (p->ev_join[k]) (void)hipEventDestroy(p->ev_join[k]);
  }
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  delete p;
}

int mf_plan_create(MfPlan** out, const Geo& G, bool keep) {
  *out = nullptr;
  MfPlan* P = new MfPlan();
  P->G = G;
  P->keep = keep;
  const int dim = G.dim, n = G.n, bs = G.bs, nn = G.nn;
  TreeBuilder tb;
  tb.dim = dim;
  tb.n = n;
  // leaves of about 64 - 81 unknowns (a leaf front pads its eliminated block to a multiple of 32): 27 nodes x 3 unknowns for 3D elasticity
  // (12 and 64 nodes measure the same within 1 %), 64 nodes for scalar 3D problems (16^3: +10 % over 9-node leaves), 32 for two unknowns
  // per node (16: the same); scalar 2D: 25 (against 64: 80^2 +10 %, 96^2 +6 %, 128^2 +3 %; 16: -30 %)
  tb.leaf_max = bs >= 3 ? 27 : bs == 2 ? 32 : dim == 2 ? 25 : 64;
  if (const char* e = getenv("HOMMX_MF_LEAF")) tb.leaf_max = std::max(1, atoi(e));
  if (const char* e = getenv("HOMMX_MF_SPLIT_DEPTH")) tb.split_depth = atoi(e);
  if (const char* e = getenv("HOMMX_MF_STAGE")) P->stage = atoi(e);
  if (const char* e = getenv("HOMMX_MF_STREAMS")) P->streams = atoi(e);
  if (const char* e = getenv("HOMMX_MF_FRONT")) P->front_max_t = std::max(0, std::min(MFF_MAX_T, atoi(e)));
  if (keep) P->front_max_t = 0;  // the back substitution reads N_i and X_i of every front: the corrector plan keeps them in HBM
  {
    std::vector<int> all(nn);
    for (int i = 0; i < nn; ++i) all[i] = i;
    const int lo[3] = {0, 0, 0}, hi[3] = {n, n, dim == 3 ? n : 1};
    const bool per[3] = {true, true, dim == 3};
    tb.rec(all, lo, hi, per);
  }
  std::vector<SN>& sn = tb.sn;
  const int nsn = (int)sn.size();
  // elimination rank of every node: supernodes in list order (children before parents), ascending node id inside
  std::vector<int> rank(nn, -1), owner(nn, -1);
  {
    int r = 0;
    for (int k = 0; k < nsn; ++k) {
      std::sort(sn[k].nodes.begin(), sn[k].nodes.end());
      for (int v : sn[k].nodes) {
        rank[v] = r++;
        owner[v] = k;
      }
    }
  }
  // stencil offsets: code = sum (d_k + 1) 3^k; the periodic P1 stencil couples along the cube's main-diagonal tetrahedra only
  // (offsets whose components all have one sign), but the table keeps every code the assembly kernel writes (zeros included)
  const int ncode = G.ncode;
  auto neighbour = [&](int v, int code) {
    int out = 0, mul = 1, vv = v, cc = code;
    for (int k = 0; k < dim; ++k) {
      const int c = vv % n, d = cc % 3 - 1;
      out += ((c + d + n) % n) * mul;
      mul *= n;
      vv /= n;
      cc /= 3;
    }
    return out;
  };
  auto coupled = [&](int code) {  // offsets of the 7- / 15-point stencil (all components >= 0 or all <= 0)
    bool pos = false, neg = false;
    int cc = code;
    for (int k = 0; k < dim; ++k) {
      const int d = cc % 3 - 1;
      pos |= d > 0;
      neg |= d < 0;
      cc /= 3;
    }
    return !(pos && neg);
  };
  // boundaries (symbolic elimination on the supernode tree) and heights
  {
    std::vector<char> mark(nn, 0);
    for (int k = 0; k < nsn; ++k) {
      SN& s = sn[k];
      std::vector<int> cand;
      for (int v : s.nodes)
        for (int code = 0; code < ncode; ++code)
          if (coupled(code)) cand.push_back(neighbour(v, code));
      for (int c : s.children) {
        cand.insert(cand.end(), sn[c].bnd.begin(), sn[c].bnd.end());
        s.height = std::max(s.height, sn[c].height + 1);
      }
      for (int v : cand)
        if (owner[v] > k && !mark[v]) {
          mark[v] = 1;
          s.bnd.push_back(v);
        }
      for (int v : s.bnd) mark[v] = 0;
      std::sort(s.bnd.begin(), s.bnd.end(), [&](int a, int b) { return rank[a] < rank[b]; });
    }
  }
  if (owner[nn - 1] != nsn - 1) {
    g_berr = "multifrontal plan: the gauge node is not in the root front";
    delete P;
    return HOMMX_EINVAL;
  }
  // groups: fronts of equal (height, ns, nr)
  std::map<std::tuple<int, int, int>, int> gid;
  std::vector<std::vector<int>> members;
  for (int k = 0; k < nsn; ++k) {
    auto key = std::make_tuple(sn[k].height, (int)sn[k].nodes.size(), (int)sn[k].bnd.size());
    auto it = gid.find(key);
    if (it == gid.end()) {
      it = gid.emplace(key, (int)members.size()).first;
      members.emplace_back();
    }
    sn[k].group = it->second;
    sn[k].fidx = (int)members[it->second].size();
    members[it->second].push_back(k);
  }
  const int ng = (int)members.size();  // std::map iteration order = (height, ns, nr) ascending = a valid processing order
  std::vector<int> order;
  for (auto& kv : gid) order.push_back(kv.second);
  std::vector<int> pos_of(ng);
  for (int t = 0; t < ng; ++t) pos_of[order[t]] = t;
  P->groups.resize(ng);
  P->nfronts = nsn;
  // sizes, lifetimes, arena offsets (first fit over the live intervals)
  std::vector<int> expiry(ng);
  for (int t = 0; t < ng; ++t) {
    const int g = order[t];
    MfGroup& mg = P->groups[t];
    const SN& s0 = sn[members[g][0]];
    mg.height = s0.height;
    mg.ns = (int)s0.nodes.size();
    mg.nr = (int)s0.bnd.size();
    mg.sp = round_up(mg.ns * bs, 32);
    mg.rb = mg.nr * bs;
    mg.rp = round_up(mg.rb + MF_BORDER, 16);
    mg.L = mg.sp + mg.rp;
    mg.nf = (int)members[g].size();
    mg.s16 = round_up(mg.ns * bs, 16);
    mg.T = (mg.s16 + round_up(mg.rb + MF_BORDER, 16)) / 16;
    mg.P = mg.s16 / 16;
    mg.ntiles = mg.T * (mg.T + 1) / 2;
    mg.R0 = 0;
    mg.nreg = mg.ntiles;
    while (mg.nreg > MFF_REG_TILES) {  // the first tile rows move to LDS until the rest fits the registers of eight waves
      mg.nreg -= mg.T - mg.R0;
      ++mg.R0;
    }
    mg.has_children = !s0.children.empty();
    // the root front pins the gauge node (k_mf_pad does that on the launch sequence): it stays there
    mg.front = mg.T <= P->front_max_t && mg.R0 <= 2 && mg.R0 <= mg.P && members[g][0] != nsn - 1 && s0.height < sn[nsn - 1].height;
    expiry[t] = t;
  }
  for (int k = 0; k < nsn; ++k)
    for (int c : sn[k].children) expiry[pos_of[sn[c].group]] = std::max(expiry[pos_of[sn[c].group]], pos_of[sn[k].group]);
  if (keep)
    for (int t = 0; t < ng; ++t) expiry[t] = ng;  // nothing is ever released
  {
    struct Live {
      long long off, size;
      int expiry;
    };
    std::vector<Live> live;
    long long peak = 0;
    auto place = [&](long long size, int exp_, int now) {
      live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.expiry < now; }), live.end());
      std::sort(live.begin(), live.end(), [](const Live& a, const Live& b) { return a.off < b.off; });
      long long off = 0;
      for (const Live& l : live) {
        if (off + size <= l.off) break;
        off = std::max(off, l.off + l.size);
      }
      live.push_back({off, size, exp_});
      peak = std::max(peak, off + size);
      return off;
    };
    for (int t = 0; t < ng; ++t) {
      MfGroup& mg = P->groups[t];
      mg.offF = place((long long)mg.nf * mg.L * mg.L, expiry[t], t);
      P->scratch_per_cell = std::max(P->scratch_per_cell, (long long)mg.nf * mg.sp * mg.sp);
      if (keep) P->vbuf_per_cell = std::max(P->vbuf_per_cell, (long long)mg.nf * MF_BORDER * mg.L);
      if (mg.front) {  // k_mf_front: per panel p of 16 pivots the sweep, T - 1 - p products Y'_a and (T - 1 - p)(T - p) / 2 tile updates of 4 MFMAs
        double f = 0.0;
        for (int pp = 0; pp < mg.P; ++pp) {
          const double rest = mg.T - 1 - pp;
          f += 2.0 * 16 * 16 * 16 + 8192.0 * (rest + rest * (rest + 1) / 2);
        }
        P->flops_per_cell += mg.nf * f;
      } else {  // inverse of a stage si^3, X_i 2 si^2 below, column update 2 below rem si, Schur update s r^2 (lower tiles)
        const int nst = mf_stages(mg.sp, P->stage);
        double f = (double)mg.sp * mg.rp * mg.rp;
        int off = 0;
        for (int i = 0; i < nst; ++i) {
          const double si = mf_stage_size(mg.sp, nst, i), below = mg.L - (off + si), rem = mg.sp - (off + si);
          f += si * si * si + 2.0 * si * si * below + 2.0 * below * rem * si;
          off += (int)si;
        }
        P->flops_per_cell += mg.nf * f;
      }
    }
    P->arena_per_cell = peak;
  }
  // device tables
  std::vector<int> local(nn, -1);
  for (int t = 0; t < ng; ++t) {
    const int g = order[t];
    MfGroup& mg = P->groups[t];
    const int nloc = mg.ns + mg.nr;
    std::vector<int32_t> nodes((size_t)mg.nf * nloc), cpos((size_t)mg.nf * 2 * nloc, -1), dpos((size_t)mg.nf * 2 * mg.rp, -1);
    std::vector<int8_t> code((size_t)mg.nf * nloc * mg.ns + 4, (int8_t)-1);  // + 4: k_mf_front stages the tables word by word
    std::vector<MfChild> child((size_t)mg.nf * 2);
    const int NUf = 16 * mg.T;
    std::vector<int32_t> upos(mg.front ? (size_t)mg.nf * 2 * NUf : 0, -1);
    for (int f = 0; f < mg.nf; ++f) {
      const SN& s = sn[members[g][f]];
      int32_t* nd = &nodes[(size_t)f * nloc];
      for (int i = 0; i < mg.ns; ++i) nd[i] = s.nodes[i];
      for (int i = 0; i < mg.nr; ++i) nd[mg.ns + i] = s.bnd[i];
      for (int i = 0; i < nloc; ++i) local[nd[i]] = i;
      for (int j = 0; j < mg.ns; ++j)  // column node j: its stencil neighbours i = j + off(code'), stored as the code of (i -> j)
        for (int c = 0; c < ncode; ++c) {
          if (!coupled(c)) continue;
          const int v = neighbour(nd[j], c);  // v = node_j + off(c)  =>  node_j = v + off(opposite code)
          const int i = local[v];
          if (i < 0) continue;  // eliminated earlier: that coupling sits in an earlier front
          code[((size_t)f * nloc + i) * mg.ns + j] = (int8_t)(ncode - 1 - c);  // opposite offset: code of (node_i -> node_j)
        }
      for (int slot = 0; slot < 2; ++slot) {
        MfChild& ch = child[(size_t)f * 2 + slot];
        ch = MfChild{0, 0, 0, 0, 0, 0, 0};
        if (slot >= (int)s.children.size()) continue;
        const SN& cs = sn[s.children[slot]];
        const MfGroup& cg = P->groups[pos_of[cs.group]];
        ch.offF = cg.offF;
        ch.nf = cg.nf;
        ch.fidx = cs.fidx;
        ch.L = cg.L;
        ch.sp = cg.sp;
        ch.rb = cg.rb;
        ch.valid = 1;
        for (int q = 0; q < (int)cs.bnd.size(); ++q) {
          const int i = local[cs.bnd[q]];
          if (i < 0) {
            g_berr = "multifrontal plan: a child's boundary node is missing from its parent's front";
            mf_plan_destroy(P);
            return HOMMX_EINVAL;
          }
          cpos[((size_t)f * 2 + slot) * nloc + i] = q;
        }
        int32_t* dp = &dpos[((size_t)f * 2 + slot) * mg.rp];
        for (int p = 0; p < mg.rb; ++p) {
          const int q = cpos[((size_t)f * 2 + slot) * nloc + mg.ns + p / bs];
          dp[p] = q < 0 ? -1 : q * bs + p % bs;
        }
        for (int m = 0; m < MF_BORDER; ++m) dp[mg.rb + m] = cg.rb + m;  // border row m of the child -> border row m here
        if (mg.front) {  // the same maps per UNKNOWN of the 16-granular front: eliminated unknowns, boundary unknowns, border rows
          int32_t* up = &upos[((size_t)f * 2 + slot) * NUf];
          for (int u = 0; u < mg.ns * bs; ++u) {
            const int q = cpos[((size_t)f * 2 + slot) * nloc + u / bs];
            up[u] = q < 0 ? -1 : q * bs + u % bs;
          }
          for (int p = 0; p < mg.rb + MF_BORDER; ++p) up[mg.s16 + p] = dp[p];
        }
      }
      if (s.children.size() > 2) {
        g_berr = "multifrontal plan: more than two children";
        mf_plan_destroy(P);
        return HOMMX_EINVAL;
      }
      if (members[g][f] == nsn - 1) mg.pinpos = local[nn - 1];
      for (int i = 0; i < nloc; ++i) local[nd[i]] = -1;
    }
    std::vector<uint16_t> tilemap;
    if (mg.front)
      for (int a = mg.R0; a < mg.T; ++a)   // row by row: the tiles with eliminated rows (a < P) come first (mf_front_kernel.h; the one-wave variants,
        for (int b = a; b < mg.T; ++b) tilemap.push_back((uint16_t)(a << 8 | b));  // T <= 6, number their tiles column by column without a table)
    if (upload(&mg.d_nodes, nodes) || upload(&mg.d_code, code) || upload(&mg.d_cpos, cpos) || upload(&mg.d_dpos, dpos) ||
        upload(&mg.d_child, child) || upload(&mg.d_upos, upos) || upload(&mg.d_tilemap, tilemap)) {
      mf_plan_destroy(P);
      return HOMMX_EHIP;
    }
  }
  if (getenv("HOMMX_MF_VERBOSE")) {
    fprintf(stderr, "[hommx multifrontal] n = %d, bs = %d: %d fronts in %d groups, arena %.1f MB + scratch %.1f MB per cell, %.2f GFLOP per cell\n",
            n, bs, nsn, ng, 8e-6 * P->arena_per_cell, 8e-6 * P->scratch_per_cell, 1e-9 * P->flops_per_cell);
    for (const MfGroup& mg : P->groups)
      fprintf(stderr, "   height %d: %3d fronts  s = %4d (%4d)  r = %4d (%4d)%s\n", mg.height, mg.nf, mg.ns * bs, mg.sp, mg.rb, mg.rp,
              mg.front ? "  [k_mf_front]" : "");
  }
  *out = P;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------
struct MfGroupDev {
  int ns, nloc, sp, rb, L, nf;
  long long offF;         // per-cell arena offset of this group's fronts
  const int32_t* nodes;
  const int8_t* code;
  const int32_t* cpos;
  const MfChild* child;
};

// The eliminated COLUMNS of every (cell, front) of the group -- F11 (complete) and F21 with its border rows: original stencil entries +
// the children's update matrices (+ the canonical loads in the border rows).  One thread per (row node i or the border, column UNKNOWN
// q = j BS + b of an eliminated node j): BS values (or the MF_BORDER border entries) of one column, so that the 64 lanes of a wave
// write -- and read from the children -- contiguous row segments.  F22 is not built (see the header comment).
template <int BS>
__global__ __launch_bounds__(256) void k_mf_build(MfGroupDev g, const double* __restrict__ Kst, const double* __restrict__ Brhs,
                                                  double* __restrict__ arena, long long nc, int nn, int ncode, int t, int jblocks,
                                                  int pshift, long long nblocks) {
  constexpr int RPT = MF_BUILD_ROWS / 4;  // rows per thread
  const int tid = threadIdx.x;
  // narrow fronts (ns * BS <= 32 column unknowns) pack 2^pshift rows into one wave: lane = (row within the pack, column unknown)
  const int rows_blk = MF_BUILD_ROWS << pshift;
  const int iblocks = (g.nloc + 1 + rows_blk - 1) / rows_blk;
  // a launch holds at most 2^32 - 1 work-items (AQL grid size): big batches walk the block index with a grid stride
  for (long long blk0 = blockIdx.x; blk0 < nblocks; blk0 += gridDim.x) {
    long long blk = blk0;
    const int jb = (int)(blk % jblocks);
    blk /= jblocks;
    const int ib = (int)(blk % iblocks);
    const long long batch = blk / iblocks;
    const long long cell = batch / g.nf;
    const int f = (int)(batch % g.nf);
    const int32_t* nodes = g.nodes + (long long)f * g.nloc;
    double* F = arena + nc * g.offF + batch * (long long)g.L * g.L;
    const int lpr = 64 >> pshift;  // lanes per row
    const int q = jb * 64 + (tid & (lpr - 1));
    const int j = q / BS, b = q - j * BS;
    if (j >= g.ns) continue;
    // the two child slots of the front and where column node j sits in their boundary lists
    const MfChild ch0 = g.child[f * 2], ch1 = g.child[f * 2 + 1];
    const int32_t* cp0 = g.cpos + ((long long)f * 2) * g.nloc;
    const int32_t* cp1 = cp0 + g.nloc;
    const int cj0 = ch0.valid ? cp0[j] : -1, cj1 = ch1.valid ? cp1[j] : -1;
    const double* U0 = arena + nc * ch0.offF + ((cell * ch0.nf + ch0.fidx) * (long long)ch0.L + ch0.sp) * ch0.L + ch0.sp;
    const double* U1 = arena + nc * ch1.offF + ((cell * ch1.nf + ch1.fidx) * (long long)ch1.L + ch1.sp) * ch1.L + ch1.sp;
    // RPT rows per thread (i0, i0 + 4, ...), in three passes -- indices, values, stores -- so that the dependent loads of all rows are in
    // flight together (one row at a time the kernel is bound by load latency, not by traffic)
    const int i0 = ib * rows_blk + ((tid >> 6) << pshift) + ((tid & 63) >> (6 - pshift));
    const int istep = 4 << pshift;
    int cd[RPT], c0[RPT], c1[RPT], nd[RPT];
    bool ok[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int i = i0 + istep * r;
      ok[r] = i < g.nloc && !(i < g.ns && j > i);
      cd[r] = ok[r] ? (int)g.code[((long long)f * g.nloc + i) * g.ns + j] : -1;
      c0[r] = (ok[r] && cj0 >= 0) ? cp0[i] : -1;
      c1[r] = (ok[r] && cj1 >= 0) ? cp1[i] : -1;
      nd[r] = ok[r] ? nodes[i] : 0;
    }
    double v[RPT][BS];  // v[r][a] = F[(i_r, a)][(j, b)]
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
#pragma unroll
      for (int a = 0; a < BS; ++a) v[r][a] = 0.0;
      if (cd[r] >= 0) {
        const double* Kc = Kst + ((cell * ncode + cd[r]) * BS) * BS * (long long)nn + nd[r];
#pragma unroll
        for (int a = 0; a < BS; ++a) v[r][a] = Kc[((long long)a * BS + b) * nn];
      }
      // only entries on and below the diagonal of a child's update matrix are valid (its GEMM updates lower TILES, and a BS x BS diagonal
      // block may straddle a tile boundary): a diagonal block is read through its lower triangle
      if (c0[r] >= 0) {
        const bool diag = c0[r] == cj0;
#pragma unroll
        for (int a = 0; a < BS; ++a) {
          const int ra = (diag && b > a) ? b : a, rb = (diag && b > a) ? a : b;
          v[r][a] += U0[(long long)(c0[r] * BS + ra) * ch0.L + cj0 * BS + rb];
        }
      }
      if (c1[r] >= 0) {
        const bool diag = c1[r] == cj1;
#pragma unroll
        for (int a = 0; a < BS; ++a) {
          const int ra = (diag && b > a) ? b : a, rb = (diag && b > a) ? a : b;
          v[r][a] += U1[(long long)(c1[r] * BS + ra) * ch1.L + cj1 * BS + rb];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if (!ok[r]) continue;
      const int i = i0 + istep * r;
      const int ri = i < g.ns ? i * BS : g.sp + (i - g.ns) * BS;
#pragma unroll
      for (int a = 0; a < BS; ++a) F[(long long)(ri + a) * g.L + q] = v[r][a];
      if (i < g.ns && i != j) {  // F11 is kept complete (the leaf inverses read whole diagonal blocks)
#pragma unroll
        for (int a = 0; a < BS; ++a) F[(long long)q * g.L + ri + a] = v[r][a];
      }
    }
    // border rows (load case m against unknown q): the thread row that reaches i == nloc
    if (i0 <= g.nloc && g.nloc < i0 + istep * RPT && (g.nloc - i0) % istep == 0) {
      double w[MF_BORDER];
#pragma unroll
      for (int m = 0; m < MF_BORDER; ++m) w[m] = m < t ? Brhs[cell * (long long)t * BS * nn + ((long long)m * BS + b) * nn + nodes[j]] : 0.0;
      if (cj0 >= 0) {
#pragma unroll
        for (int m = 0; m < MF_BORDER; ++m) w[m] += U0[(long long)(ch0.rb + m) * ch0.L + cj0 * BS + b];
      }
      if (cj1 >= 0) {
#pragma unroll
        for (int m = 0; m < MF_BORDER; ++m) w[m] += U1[(long long)(ch1.rb + m) * ch1.L + cj1 * BS + b];
      }
#pragma unroll
      for (int m = 0; m < MF_BORDER; ++m) F[(long long)(g.sp + g.rb + m) * g.L + q] = w[m];
    }
  }
}

// padding of the eliminated block (identity) and of the boundary rows behind the border (zeros); root: the gauge node's unknowns are pinned.
// Work items per front, each a CONTIGUOUS run along a row (the padding columns of one row are adjacent: written as one segment, not as
// npad stride-L column walks):  rows [0, nrow) x the npad_s padding columns | the npad_s padding rows x [0, sp) | the npad_r rows behind
// the border x [0, sp) | (root) 2 BS pin lines.
template <int BS>
__global__ __launch_bounds__(256) void k_mf_pad(MfGroupDev g, double* __restrict__ arena, long long nc, int pinpos) {
  const int s0 = g.ns * BS, npad_s = g.sp - s0, nrow = g.sp + g.rb + MF_BORDER, npad_r = g.L - nrow, npin = pinpos >= 0 ? BS : 0;
  const int wA = nrow * npad_s, wB = npad_s * g.sp, wC = npad_r * g.sp, wD = 2 * npin * nrow, per = wA + wB + wC + wD;
  const long long nb = nc * g.nf;
  for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {  // one workgroup per front: 32-bit index arithmetic inside
    double* F = arena + nc * g.offF + batch * (long long)g.L * g.L;
    for (int w0 = threadIdx.x; w0 < per; w0 += 256) {
      int w = w0;
      if (w < wA) {  // column part of the identity padding: F[x][s0 + q], x < nrow
        const int x = w / npad_s, p = s0 + w % npad_s;
        F[(long long)x * g.L + p] = x == p ? 1.0 : 0.0;
      } else if ((w -= wA) < wB) {  // row part: F[s0 + q][x], x < sp
        const int p = s0 + w / g.sp, x = w % g.sp;
        F[(long long)p * g.L + x] = x == p ? 1.0 : 0.0;
      } else if ((w -= wB) < wC) {  // F21 rows behind the border
        const int p = nrow + w / g.sp, x = w % g.sp;
        F[(long long)p * g.L + x] = 0.0;
      } else {  // gauge: unit row and column of the pinned unknowns (the load rows lose their entry there too)
        w -= wC;
        const int qq = w / (2 * nrow), y = w % (2 * nrow), p = pinpos * BS + qq;
        if (y < nrow) F[(long long)y * g.L + p] = y == p ? 1.0 : 0.0;
        else if (y - nrow < g.sp) F[(long long)p * g.L + (y - nrow)] = (y - nrow) == p ? 1.0 : 0.0;
      }
    }
  }
}

// A_H = C0 + corner of the root front's update matrix (= -B^T K^+ B, lower triangle valid)
__global__ void k_mf_finalize(const double* __restrict__ C0, const double* __restrict__ arena, long long offF, int L, int sp, int rb, int t,
                              double* __restrict__ out, long long nc) {
  const int tt = t * t;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nc * tt) return;
  const long long cell = idx / tt;
  const int m = (int)(idx % tt) / t, q = (int)(idx % tt) % t;
  const int hi = m > q ? m : q, lo = m > q ? q : m;
  const double* F = arena + nc * offF + cell * (long long)L * L;
  out[idx] = C0[idx] + F[(long long)(sp + rb + hi) * L + sp + rb + lo];
}

// ---------------------------------------------------------------------------------------------------------------
// correctors: back substitution down the tree (corrector plan: every front still holds N_i and X_i of its stages)
//   per front and load case m, with v = the solution at the front's unknowns in front order:
//     v[boundary]  = the solution found by the ancestors (gathered from the output array)
//     v[stage i]   = X_i[:, border m] - X_i[:, later stages and boundary] v[later stages and boundary]     (last stage first)
//   X_i = N_i E_i^T, and border column m of X_i is N_i times the forward-eliminated load vector (the border rows took part in the
//   elimination).  The products run as thin batched GEMMs (k_gemm_tile, M = 8 load rows); V is [front][8][L].
// ---------------------------------------------------------------------------------------------------------------
template <int BS>
__global__ __launch_bounds__(256) void k_mf_bs_init(MfGroupDev g, const double* __restrict__ arena, double* __restrict__ V,
                                                    const double* __restrict__ corr, long long nc, int t, long long ndof) {
  const long long nb = nc * g.nf;
  for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
    const long long cell = batch / g.nf;
    const int f = (int)(batch % g.nf);
    const int32_t* nodes = g.nodes + (long long)f * g.nloc;
    const double* F = arena + nc * g.offF + batch * (long long)g.L * g.L;
    double* v = V + batch * (long long)MF_BORDER * g.L;
    for (int w = threadIdx.x; w < MF_BORDER * g.L; w += 256) {
      const int m = w / g.L, q = w % g.L;
      double x = 0.0;
      if (q < g.sp) x = F[(long long)q * g.L + g.sp + g.rb + m];
      else if (q < g.sp + g.rb && m < t) {
        const int p = q - g.sp;
        x = corr[(cell * t + m) * ndof + (long long)nodes[g.ns + p / BS] * BS + p % BS];
      }
      v[w] = x;
    }
  }
}

template <int BS>
__global__ __launch_bounds__(256) void k_mf_bs_store(MfGroupDev g, const double* __restrict__ V, double* __restrict__ corr, long long nc,
                                                     int t, long long ndof) {
  const long long nb = nc * g.nf;
  const int s0 = g.ns * BS;
  for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
    const long long cell = batch / g.nf;
    const int f = (int)(batch % g.nf);
    const int32_t* nodes = g.nodes + (long long)f * g.nloc;
    const double* v = V + batch * (long long)MF_BORDER * g.L;
    for (int w = threadIdx.x; w < t * s0; w += 256) {
      const int m = w / s0, q = w % s0;
      corr[(cell * t + m) * ndof + (long long)nodes[q / BS] * BS + q % BS] = v[m * g.L + q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------------------
int mf_reserve(BlockedWorkspace* ws, MfPlan* P, long long ncells, bool ahead) {
  const Geo& G = ws->G;
  const long long stencil = (long long)G.ncode * G.bs * G.bs * G.nn + (long long)G.t * G.bs * G.nn + 36;
  const long long per_cell = 8ll * (P->arena_per_cell + P->scratch_per_cell + P->vbuf_per_cell + stencil);
  // Fronts are big (C4 / C5: 0.2 GB per cell) and the card has 288 GB, but a call pays for what it allocates (2.5 - 3 s per 64 GB
  // measured) and the throughput is nearly flat from 256 cells per chunk (C4: 2,390 / 2,510 / 2,580 / 2,655 / 2,654 solves/s at 64 / 128 /
  // 192 / 256 / 384 cells; all of C5: 2,590 with 64 GB, 2,640 with 128).  A solve that has to allocate takes 64 GB (more, to 128 GB, only
  // where 256 cells need it); hommx_plan_reserve -- the caller pays ahead of many batches -- takes 128 GB.
  double budget_gb = ahead ? 128.0 : std::max(64.0, std::min(128.0, 256.0 * 1e-9 * (double)per_cell));
  {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
      const double have = 1e-9 * (double)fr + 8e-9 * (double)P->chunk * (P->arena_per_cell + P->scratch_per_cell);  // what is free + what we hold
      budget_gb = std::min(budget_gb, 0.6 * have);
    }
  }
  if (ws->budget_gb_env > 0.0) budget_gb = ws->budget_gb_env;
  long long chunk = (long long)(budget_gb * 1e9) / per_cell;
  if (chunk < 1) chunk = 1;
  if (chunk > 4096) chunk = 4096;
  if (chunk > ncells) chunk = ncells;
  // tile orders of the gathering Schur updates of every group, made here: no allocation and no blocking copy later, while the streams fill
  if (ws->tile_sb > 1)
    for (const MfGroup& mg : P->groups)
      for (int mn : {mg.rb, mg.rp}) {
        const int TM = gemm_tile_size(ws, mn, mn, mg.sp, true), ty = (mn + TM - 1) / TM;
        if (mn > 0 && ty >= 2 * ws->tile_sb) (void)ensure_tilemap(ws, ty);
      }
  if (chunk <= P->chunk) return 0;
  for (double** p : {&P->arena, &P->scratch, &P->vbuf, &P->Kst, &P->Brhs, &P->C0}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  P->chunk = 0;
  MTRY(hipMalloc(&P->arena, 8ll * chunk * P->arena_per_cell));
  MTRY(hipMalloc(&P->scratch, 8ll * chunk * P->scratch_per_cell));
  if (P->vbuf_per_cell) MTRY(hipMalloc(&P->vbuf, 8ll * chunk * P->vbuf_per_cell));
  MTRY(hipMalloc(&P->Kst, 8ll * chunk * G.ncode * G.bs * G.bs * G.nn));
  MTRY(hipMalloc(&P->Brhs, 8ll * chunk * G.t * G.bs * G.nn));
  MTRY(hipMalloc(&P->C0, 8ll * chunk * 36));
  P->chunk = chunk;
  return 0;
}

static inline unsigned nblk(long long work, int bs = 256) { return (unsigned)((work + bs - 1) / bs); }

namespace {

// one half of a chunk: its cells, where its buffers start inside the chunk buffers (in cells) and the stream it runs on
struct MfHalf {
  long long c0, nc, base;
  hipStream_t st;
};

// all launches of one group of fronts for one half
void mf_group_step(BlockedWorkspace* ws, MfPlan* P, const MfHalf& h, const MfGroup& mg, int gi, int32_t* d_info) {
  const Geo& G = ws->G;
  const int bs = G.bs;
  const long long nc = h.nc;
  hipStream_t st = h.st;
  double* arena = P->arena + h.base * P->arena_per_cell;
  double* scratch = P->scratch + h.base * P->scratch_per_cell;
  const double* Kst = P->Kst + h.base * G.ncode * bs * bs * G.nn;
  const double* Brhs = P->Brhs + h.base * G.t * bs * G.nn;
  const long long nb = nc * mg.nf;  // matrices in this batch
  if (mg.front) {  // the whole group in ONE launch: fronts live in registers (mf_front.h)
    MfFrontDev fd{mg.ns, mg.ns + mg.nr, mg.sp, mg.rb, mg.L, mg.nf, mg.s16, mg.T, mg.P, mg.nreg, mg.has_children ? 1 : 0, mg.R0, mg.offF,
                  mg.d_nodes, mg.d_code, mg.d_upos, mg.d_child, mg.d_tilemap};
    launch_mf_front(fd, bs, Kst, Brhs, arena, nc, nb, G.nn, G.ncode, G.t, d_info ? d_info + h.c0 : nullptr, gi, st);
    return;
  }
  MfGroupDev gd{mg.ns, mg.ns + mg.nr, mg.sp, mg.rb, mg.L, mg.nf, mg.offF, mg.d_nodes, mg.d_code, mg.d_cpos, mg.d_child};
  const int jblocks = (mg.ns * bs + 63) / 64;  // 64 column unknowns per wave
  int pshift = 0;  // rows of a narrow front packed into one wave (k_mf_build)
  while (pshift < 3 && mg.ns * bs * (2 << pshift) <= 64) ++pshift;
  const int rows_blk = MF_BUILD_ROWS << pshift;
  const long long bblocks = nb * ((gd.nloc + 1 + rows_blk - 1) / rows_blk) * jblocks;
  const long long max_blocks = 1ll << 22;  // x 256 threads = 2^30 work-items per launch (the AQL limit is 2^32 - 1); grid-stride beyond
  const long long pad_work = (long long)(mg.sp + mg.rb + MF_BORDER) * (mg.sp - mg.ns * bs) + (long long)(mg.sp - mg.ns * bs) * mg.sp +
                             (long long)(mg.rp - mg.rb - MF_BORDER) * mg.sp + (mg.pinpos >= 0 ? 2ll * bs * (mg.sp + mg.rb + MF_BORDER) : 0);
#define HOMMX_MF_K(BS_)                                                                                                            \
  do {                                                                                                                             \
    hipLaunchKernelGGL((k_mf_build<BS_>), dim3((unsigned)std::min(bblocks, max_blocks)), dim3(256), 0, st, gd, Kst, Brhs, arena,   \
                       nc, G.nn, G.ncode, G.t, jblocks, pshift, bblocks);                                                          \
    if (pad_work > 0)                                                                                                              \
      hipLaunchKernelGGL((k_mf_pad<BS_>), dim3((unsigned)std::min(nb, max_blocks)), dim3(256), 0, st, gd, arena, nc, mg.pinpos);   \
  } while (0)
  if (bs == 1) HOMMX_MF_K(1);
  else if (bs == 2) HOMMX_MF_K(2);
  else HOMMX_MF_K(3);
#undef HOMMX_MF_K
  double* F = arena + nc * mg.offF;
  const long long sF = (long long)mg.L * mg.L;
  Ctx c{ws, nb, st, d_info ? d_info + h.c0 : nullptr, gi};
  c.ld = mg.L;
  c.sS = sF;
  c.sT = (long long)mg.sp * mg.sp;
  c.infoDiv = mg.nf;
  // Elimination of the s unknowns in STAGES of about `stage` unknowns (one stage for small fronts): stage i inverts its diagonal
  // block, N_i, forms X_i = N_i E_i^T for every row below it (E_i: the rows below, columns of the stage) into the free upper part
  // of the front, and updates the not yet eliminated COLUMNS only.  After the last stage the rows of F12 hold [X_1; X_2; ...]
  // restricted to the boundary columns and F21 holds the updated [E_1 E_2' ...]: the Schur update below is one product of rank s
  // all the same, but the work in front of it drops from s^3 + 2 s^2 r to about 0.75 s^3 + 1.5 s^2 r (two stages).
  double* F21 = F + (long long)mg.sp * mg.L;
  double* F12 = F + mg.sp;
  double* F22 = F21 + mg.sp;
  {
    const int nst = mf_stages(mg.sp, P->stage);
    int off = 0;
    for (int i = 0; i < nst; ++i) {
      const int si = mf_stage_size(mg.sp, nst, i);
      invert(c, F, off, si, scratch);                                                        // N_i
      const int below = mg.L - (off + si), rem = mg.sp - (off + si);
      double* Ni = F + (long long)off * mg.L + off;
      double* Ei = F + (long long)(off + si) * mg.L + off;       // rows below the stage, its columns
      double* Xi = F + (long long)off * mg.L + off + si;         // si x below, in the upper part of the front
      gemm(c, true, true, si, below, si, 1.0, Ni, mg.L, sF, Ei, mg.L, sF, 0.0, Xi, mg.L, sF);   // X_i = N_i E_i^T (N_i symmetric: read as its
                                                                                               //  transpose, the k-major staging path)
      if (rem > 0)                                                                            // columns still to eliminate -= E_i X_i
        gemm(c, false, false, below, rem, si, -1.0, Ei, mg.L, sF, Xi, mg.L, sF, 1.0, F + (long long)(off + si) * mg.L + off + si, mg.L, sF);
      off += si;
    }
  }
  GatherC ga;
  ga.arena = arena;
  ga.nc = nc;
  ga.child = mg.d_child;
  ga.dpos = mg.d_dpos;
  ga.nf = mg.nf;
  ga.rp = mg.rp;
  // F22 = children - F21 F12, lower tiles.  When the border rows would open a tile row of their own they get a (thin) launch instead.
  const int TMg = (mg.rp >= ws->gemm128_min && mg.sp >= ws->mf_gather128_min_k) ? 128 : 64;
  const bool split = mg.rb >= TMg && (mg.rp + TMg - 1) / TMg > (mg.rb + TMg - 1) / TMg && !ws->mf_no_border_split;
  const int main_n = split ? mg.rb : mg.rp;
  gemm(c, false, false, main_n, main_n, mg.sp, -1.0, F21, mg.L, sF, F12, mg.L, sF, 1.0, F22, mg.L, sF, 1, nullptr, &ga);
  if (split) {
    ga.rowOff = mg.rb;
    gemm(c, false, false, mg.rp - mg.rb, mg.rp, mg.sp, -1.0, F21 + (long long)mg.rb * mg.L, mg.L, sF, F12, mg.L, sF, 1.0,
         F22 + (long long)mg.rb * mg.L, mg.L, sF, 0, nullptr, &ga);
  }
}

// back substitution of one group of fronts for one half (corrector plan): the unknowns the group eliminates, for every load case
void mf_backsub_step(BlockedWorkspace* ws, MfPlan* P, const MfHalf& h, const MfGroup& mg, double* d_corr) {
  const Geo& G = ws->G;
  const int bs = G.bs;
  const long long nc = h.nc, nb = nc * mg.nf, ndof = (long long)G.nn * bs;
  hipStream_t st = h.st;
  double* arena = P->arena + h.base * P->arena_per_cell;
  double* V = P->vbuf + h.base * P->vbuf_per_cell;
  double* corr = d_corr + h.c0 * G.t * ndof;
  MfGroupDev gd{mg.ns, mg.ns + mg.nr, mg.sp, mg.rb, mg.L, mg.nf, mg.offF, mg.d_nodes, mg.d_code, mg.d_cpos, mg.d_child};
  const unsigned blocks = (unsigned)std::min(nb, 1ll << 22);
#define HOMMX_MF_BS(KERN, ...)                                                                           \
  do {                                                                                                   \
    if (bs == 1) hipLaunchKernelGGL((KERN<1>), dim3(blocks), dim3(256), 0, st, __VA_ARGS__);             \
    else if (bs == 2) hipLaunchKernelGGL((KERN<2>), dim3(blocks), dim3(256), 0, st, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERN<3>), dim3(blocks), dim3(256), 0, st, __VA_ARGS__);                     \
  } while (0)
  HOMMX_MF_BS(k_mf_bs_init, gd, arena, V, corr, nc, G.t, ndof);
  const double* F = arena + nc * mg.offF;
  const long long sF = (long long)mg.L * mg.L, sV = (long long)MF_BORDER * mg.L;
  Ctx c{ws, nb, st, nullptr, 0};
  const int nst = mf_stages(mg.sp, P->stage);
  int off = mg.sp;
  for (int i = nst - 1; i >= 0; --i) {  // v[stage i] -= X_i[:, later stages and boundary] v[...]: C (8 x si) -= A (8 x rest) B^T (B = X_i: si x rest)
    const int si = mf_stage_size(mg.sp, nst, i);
    off -= si;
    const int rest = mg.sp + mg.rb - (off + si);
    if (rest > 0)
      gemm(c, false, true, MF_BORDER, si, rest, -1.0, V + off + si, mg.L, sV, F + (long long)off * mg.L + off + si, mg.L, sF, 1.0, V + off,
           mg.L, sV);
  }
  HOMMX_MF_BS(k_mf_bs_store, gd, V, corr, nc, G.t, ndof);
#undef HOMMX_MF_BS
}

}  // namespace

int mf_solve(BlockedWorkspace* ws, MfPlan* P, long long ncells, const double* d_coef, const double* d_M, double* d_out, int32_t* d_info,
             hipStream_t st, double* d_corr) {
  const Geo& G = ws->G;
  if (d_corr && !P->keep) {
    g_berr = "mf_solve: correctors need the corrector plan";
    return HOMMX_EINVAL;
  }
  if (int rc = mf_reserve(ws, P, ncells, false)) return rc;
  if (d_info) MTRY(hipMemsetAsync(d_info, 0, sizeof(int32_t) * ncells, st));
  long long step_cells = P->chunk;
  {
    const long long nchunks = (ncells + step_cells - 1) / step_cells;
    step_cells = (ncells + nchunks - 1) / nchunks;
  }
  // A chunk runs as two to four pieces side by side on as many streams (each owns its share of the chunk buffers): small launches of
  // one piece fill the tails of the others', and memory-bound and matrix-core-bound waves share the CUs.  Two streams against one: 3D
  // elasticity 8^3 +11 %, scalar 3D 16^3 +8 %, 2D 128^2 +9 %, C4 +2 %; four against two (pieces of at least 128 cells): all of C5 +2 %,
  // scalar 3D 16^3 +5 %; running the second stream a few groups behind gains nothing more.  The arithmetic of a cell does not depend on
  // the batch it is in: results are bitwise the same.
  const int want = std::max(1, std::min(4, P->streams));
  const bool two = want >= 2 && step_cells >= 16;
  if (two) {  // every stream / event only if its own slot is still empty: a call that failed part-way leaks nothing on the next one
    for (int k = 0; k + 1 < want; ++k) {
      if (!P->side[k]) MTRY(hipStreamCreateWithFlags(&P->side[k], hipStreamNonBlocking));
      if (!P->ev_join[k]) MTRY(hipEventCreateWithFlags(&P->ev_join[k], hipEventDisableTiming));
    }
    if (!P->ev_fork) MTRY(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming));
  }
  // an error after the fork must not leave work on the plan-owned streams that the caller's stream never waits for
  int forked = 0;
#define MTRY_J(expr)                                                                      \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) {                                                              \
      g_berr = std::string(#expr) + ": " + hipGetErrorString(e__);                        \
      for (int k__ = 1; k__ < forked; ++k__) (void)hipStreamSynchronize(P->side[k__ - 1]); \
      return e__ == hipErrorOutOfMemory ? HOMMX_ENOMEM : HOMMX_EHIP;                      \
    }                                                                                     \
  } while (0)
  const int bs = G.bs;
  for (long long c0 = 0; c0 < ncells; c0 += step_cells) {
    const long long nc = std::min(step_cells, ncells - c0);
    MfHalf halves[4];
    int nh = 1;
    halves[0] = MfHalf{c0, nc, 0, st};
    if (two && nc >= 2) {
      nh = (int)std::max(2ll, std::min<long long>(want, nc / 128));  // pieces of at least 128 cells beyond two
      const long long per = (nc + nh - 1) / nh;
      for (int k = 0; k < nh; ++k) {
        const long long a = std::min(nc, k * per), b = std::min(nc, (k + 1) * per);
        halves[k] = MfHalf{c0 + a, b - a, a, k == 0 ? st : P->side[k - 1]};
      }
      MTRY(hipEventRecord(P->ev_fork, st));  // the side streams start behind everything queued on st (inputs, the previous chunk)
      for (int k = 1; k < nh; ++k) {
        forked = k + 1;
        MTRY_J(hipStreamWaitEvent(P->side[k - 1], P->ev_fork, 0));
      }
    }
    for (int k = 0; k < nh; ++k) {
      const MfHalf& h = halves[k];
      launch_assembly(ws, d_coef + h.c0 * G.n_el * G.ncomp, d_M ? d_M + h.c0 * G.dim * G.dim : nullptr, h.nc, h.st,
                      P->Kst + h.base * G.ncode * bs * bs * G.nn, P->Brhs + h.base * G.t * bs * G.nn, P->C0 + h.base * 36);
    }
    int gi = 0;
    for (const MfGroup& mg : P->groups) {  // launches of the pieces interleaved: all queues fill at the same pace
      ++gi;
      for (int k = 0; k < nh; ++k) mf_group_step(ws, P, halves[k], mg, gi, d_info);
    }
    const MfGroup& root = P->groups.back();
    for (int k = 0; k < nh; ++k) {
      const MfHalf& h = halves[k];
      hipLaunchKernelGGL(k_mf_finalize, dim3(nblk(h.nc * G.t * G.t)), dim3(256), 0, h.st, P->C0 + h.base * 36,
                         P->arena + h.base * P->arena_per_cell, root.offF, root.L, root.sp, root.rb, G.t, d_out + h.c0 * G.t * G.t, h.nc);
    }
    if (d_corr) {  // back substitution: root first; then the mean of every component goes (cell_problem.py:349-361, 382)
      for (auto it = P->groups.rbegin(); it != P->groups.rend(); ++it)
        for (int k = 0; k < nh; ++k) mf_backsub_step(ws, P, halves[k], *it, d_corr);
      for (int k = 0; k < nh; ++k)
        launch_center_corr(ws, d_corr + halves[k].c0 * G.t * (long long)G.nn * bs, halves[k].nc, halves[k].st);
    }
    for (int k = 1; k < nh; ++k) {  // st continues (next chunk, the caller's work) when every piece is done
      MTRY_J(hipEventRecord(P->ev_join[k - 1], P->side[k - 1]));
      MTRY_J(hipStreamWaitEvent(st, P->ev_join[k - 1], 0));
    }
    MTRY_J(hipGetLastError());
    forked = 0;
  }
#undef MTRY_J
  return 0;
}

}  // namespace hommx
