// cwr_host_builders.hpp -- the HOST-side index builders of the transport engine, free of any HIP dependency.
//
// Everything the kernels of cwr_kernels.hpp index with is built here, once per engine or once per tile schedule, as plain
// std::vector data: the symbolic J^2 (rows reachable in two face steps), the tiling of the tiled J^2 pass (per tile the distinct
// x rows it touches and the 16-bit in-tile position of every entry), the directed links between tiles, the chains along the
// flow, the per-block tile lists of a chained pass and the carry-over codes of the columns consecutive tiles share.  A bug
// here becomes an out-of-range LDS or global access in k_sq_tiled / k_sq_numeric, i.e. a GPU fault -- so this header is also
// compiled on the CPU with -fsanitize=address,undefined and -D_GLIBCXX_ASSERTIONS (bounds-checked operator[]) by
// tests/test_host_builders.py, which drives tests/host_builders/builders_main.cpp over random meshes and compares the
// results with numpy statements of the same constructions (schedule.py, tests/test_host_builders.py).
// (Round 3's one process fault was exactly such a bug, in an experimental build: profiles/r04_b_exit_fault_forensics.txt.)
//
// cwr_engine.hip includes this file and uploads the vectors; nothing here allocates device memory or launches anything.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <array>
#include <cstddef>
#include <vector>

namespace cwr {
namespace host {

constexpr int HB_N_XCD = 8;            // = cwr::N_XCD (kept separate: this header must not pull in the HIP kernels)
constexpr int HB_SQN_MAXC = 255;       // = cwr::SQN_MAXC: longest J^2 row the row-wise numeric kernel takes (8-bit slots)

// ---- symbolic J^2 -----------------------------------------------------------------------------------------------------------
// Adjacency in CSR form over the computed rows: ptr[c] .. ptr[c+1] are the (cell, face) entries of row c, nb[j] >= 0 the local
// id of the real neighbour (owned or halo), nb[j] < 0 a ghost (boundary) cell.
struct SqPattern {
  int n_sq = 0;                         // rows with a J^2 row (a prefix of the computed rows)
  int max_row = 0;                      // longest J^2 row
  bool rowwise = true;                  // every row short enough for the row-wise numeric kernel
  std::vector<int32_t> ptr2, col2;      // CSR of J^2: columns of row c in order of first discovery
  std::vector<int32_t> pair_ptr;        // first product of row c in `slots`
  std::vector<uint8_t> slots;           // per product J[c,m] J[m,k] (faces of c ascending, then faces of m ascending): slot of k in row c
  std::vector<uint8_t> fast;            // 1: k_sq_numeric may take the row through its branch-free path
};

// Rows with a J^2 row: the longest prefix of computed rows all of whose real neighbours have rows of their own (single GPU: every
// row; partitioned with halo depth s: the core and layers 1..s-2).  Returns false when that prefix does not cover the core.
inline bool symbolic_sq(int n_owned, int n_core, int max_degree, const std::vector<int32_t>& ptr, const std::vector<int32_t>& nb,
                        SqPattern& out) {
  int n = n_owned;
  for (int c = 0; c < n_owned && n == n_owned; ++c)
    for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j)
      if (nb[(size_t)j] >= n_owned) { n = c; break; }
  if (n < n_core) return false;
  out = SqPattern();
  out.n_sq = n;
  out.ptr2.assign((size_t)n + 1, 0);
  out.pair_ptr.assign((size_t)n + 1, 0);
  const size_t nnz = (size_t)ptr[(size_t)n_owned];
  out.col2.reserve(nnz * 3 + 16);
  out.slots.reserve(nnz * 4 + 16);
  std::vector<int32_t> tmp;
  for (int c = 0; c < n; ++c) {
    tmp.clear();
    for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) {
      const int m = nb[(size_t)j];
      if (m < 0) continue;
      for (int i = ptr[(size_t)m]; i < ptr[(size_t)m + 1]; ++i) {
        const int k = nb[(size_t)i];
        // columns in order of first discovery (faces ascending, then the neighbour's faces ascending): that order does
        // not depend on the local numbering, so a partitioned run sums every row exactly like the single-GPU run
        if (k < 0) continue;
        auto it = std::find(tmp.begin(), tmp.end(), k);
        if (it == tmp.end()) { tmp.push_back(k); it = tmp.end() - 1; }
        out.slots.push_back((uint8_t)std::min<size_t>(255, (size_t)(it - tmp.begin())));   // slot of every product, in order
      }
    }
    if ((int)tmp.size() > HB_SQN_MAXC) out.rowwise = false;        // such a row needs the per-entry kernel
    out.max_row = std::max(out.max_row, (int)tmp.size());
    out.col2.insert(out.col2.end(), tmp.begin(), tmp.end());
    out.ptr2[(size_t)c + 1] = (int32_t)out.col2.size();
    out.pair_ptr[(size_t)c + 1] = (int32_t)out.slots.size();
  }
  // rows k_sq_numeric may take through its branch-free path: 0 < deg <= DEG, a J^2 row of its own, at least one real neighbour,
  // and every real neighbour's row has <= DEG entries and no ghost face (so that its r-th entry is its r-th product)
  const int DEGsel = max_degree <= 4 ? 4 : (max_degree <= 6 ? 6 : 8);
  std::vector<uint8_t> ghosty((size_t)n_owned, 0);
  out.fast.assign((size_t)n, 0);
  for (int c = 0; c < n_owned; ++c)
    for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) if (nb[(size_t)j] < 0) ghosty[(size_t)c] = 1;
  for (int c = 0; c < n; ++c) {
    const int deg = ptr[(size_t)c + 1] - ptr[(size_t)c];
    bool ok = deg > 0 && deg <= DEGsel && out.ptr2[(size_t)c + 1] > out.ptr2[(size_t)c];
    bool any = false;
    for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1] && ok; ++j) {
      const int m = nb[(size_t)j];
      if (m < 0) continue;
      any = true;
      ok = (ptr[(size_t)m + 1] - ptr[(size_t)m] <= DEGsel) && !ghosty[(size_t)m];
    }
    out.fast[(size_t)c] = (ok && any) ? 1 : 0;
  }
  return true;
}

// (round 6 A/B, CWR_TCL_POWER=1) J's own pattern in the form of SqPattern, so that the tiled pass, its tiling, chains and column reuse
// serve plain Jacobi sweeps unchanged: row c = its distinct real neighbours in order of first discovery (faces ascending); faces
// between the same two cells become ONE entry (k_j_numeric sums them in face order).  No row-wise numeric tables (rowwise = false).
inline bool symbolic_j(int n_owned, int n_core, const std::vector<int32_t>& ptr, const std::vector<int32_t>& nb, SqPattern& out) {
  (void)n_core;
  out = SqPattern();
  const int n = n_owned;
  out.n_sq = n;
  out.rowwise = false;
  out.ptr2.assign((size_t)n + 1, 0);
  out.pair_ptr.assign((size_t)n + 1, 0);
  out.fast.assign((size_t)n, 0);
  std::vector<int32_t> tmp;
  for (int c = 0; c < n; ++c) {
    tmp.clear();
    for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) {
      const int m = nb[(size_t)j];
      if (m < 0) continue;
      if (std::find(tmp.begin(), tmp.end(), m) == tmp.end()) tmp.push_back(m);
    }
    out.max_row = std::max(out.max_row, (int)tmp.size());
    out.col2.insert(out.col2.end(), tmp.begin(), tmp.end());
    out.ptr2[(size_t)c + 1] = (int32_t)out.col2.size();
  }
  return true;
}

// ---- tiling of the tiled J^2 pass ---------------------------------------------------------------------------------------------
struct Tiling {
  std::vector<int32_t> trow, vptr;      // rows [trow[t], trow[t+1]) and virtual items [vptr[t], vptr[t+1]) of tile t
  std::vector<uint16_t> vtab;           // virtual items (chunks 1.. of long rows): row in tile | chunk << 8
  std::vector<int32_t> tptr, tcols;     // per tile the DISTINCT x rows its J^2 rows touch: own rows first, then the others ascending
  std::vector<uint16_t> loc2;           // per J^2 entry: position of its column in the tile's list, pre-multiplied by K
  int max_cols = 0;                     // most distinct x rows of a tile
  int cap2 = 0;                         // most J^2 entries of a tile, rounded up to even
  int ntiles() const { return (int)trow.size() - 1; }
};

// Tiles of up to `tr` lane-group slots over rows [0, n_t): a row of more than `seg` entries occupies one slot per chunk (work
// items; seg = 1 << 20, nvmax = 0: plain fixed-size tiles).  Returns false when a single row needs more slots than a tile has or
// a tile's positions do not fit 16 bits.  n_cols = rows a column id may name (the real rows of the engine).
// (round 5) col_limit / ent_limit: a window of tr rows whose tile would touch more distinct x rows or hold more J^2 entries than
// that is cut into halves (recursively) -- the tiles of the next window still start at its multiple of tr.  A rank of a partition has a
// few such windows where its replayed strips of two neighbours meet in one window (two clusters of rows, two neighbourhoods): without the
// cut the ONE heavy tile decided the kernel configuration of every tile (a 7- or 8-rows-per-lane-group prefetch: 3 resident blocks per
// CU instead of 4 -- 29 instead of 25 us per pass on a middle rank of 8 of the 1 M-cell mesh, 85 instead of 69 at 4 M:
// profiles/r05_rank_budget.txt).  heavy (optional): windows that were cut.
inline bool build_tiling(int n_t, int tr, int seg, int nvmax, int K, int n_cols, const std::vector<int32_t>& ptr2,
                         const std::vector<int32_t>& col2, Tiling& out, int col_limit = 1 << 30, int ent_limit = 1 << 30, int* heavy = nullptr) {
  out = Tiling();
  out.trow.assign(1, 0); out.vptr.assign(1, 0);
  if (heavy) *heavy = 0;
  const bool limited = col_limit < (1 << 30) || ent_limit < (1 << 30);
  std::vector<int32_t> stamp_l;
  int stamp_id = 0;
  if (limited) stamp_l.assign((size_t)n_cols, -1);
  // distinct x rows / entries of the rows [c0, c1)
  auto fits = [&](int c0, int c1) {
    if (ptr2[(size_t)c1] - ptr2[(size_t)c0] > ent_limit) return false;
    ++stamp_id;
    int cols = 0;
    for (int c = c0; c < c1; ++c) if (stamp_l[(size_t)c] != stamp_id) { stamp_l[(size_t)c] = stamp_id; ++cols; }
    for (int q = ptr2[(size_t)c0]; q < ptr2[(size_t)c1]; ++q) { const int k = col2[(size_t)q]; if (stamp_l[(size_t)k] != stamp_id) { stamp_l[(size_t)k] = stamp_id; ++cols; } }
    return cols <= col_limit;
  };
  for (int c = 0; c < n_t;) {
    int rows = 0, virt = 0;
    while (c + rows < n_t && rows < 256) {
      const int len = ptr2[(size_t)(c + rows) + 1] - ptr2[(size_t)(c + rows)];
      const int extra = (len > seg) ? (len - 1) / seg : 0;
      if (rows + 1 + virt + extra > tr || virt + extra > nvmax || extra > 255) break;
      for (int ch = 1; ch <= extra; ++ch) out.vtab.push_back((uint16_t)(rows | (ch << 8)));
      ++rows; virt += extra;
    }
    if (rows == 0) return false;                                   // a single row needs more slots than a tile has
    if (limited && nvmax == 0 && rows > 1 && !fits(c, c + rows)) {
      // cut the window: halves, and halves of those, until every piece fits (a single row that does not is the caller's problem:
      // the configuration test behind this call fails and the next larger configuration is tried)
      if (heavy) ++*heavy;
      std::vector<std::pair<int, int>> todo{{c, c + rows}}, done;
      while (!todo.empty()) {
        const auto seg_ = todo.back(); todo.pop_back();
        if (seg_.second - seg_.first <= 1 || fits(seg_.first, seg_.second)) { done.push_back(seg_); continue; }
        const int mid = (seg_.first + seg_.second) / 2;
        todo.push_back({mid, seg_.second}); todo.push_back({seg_.first, mid});
      }
      std::sort(done.begin(), done.end());
      for (const auto& d : done) { out.trow.push_back(d.second); out.vptr.push_back((int32_t)out.vtab.size()); }
      c += rows;
      continue;
    }
    c += rows;
    out.trow.push_back(c); out.vptr.push_back((int32_t)out.vtab.size());
  }
  const int nt = out.ntiles();
  out.tptr.assign((size_t)nt + 1, 0);
  out.loc2.assign((size_t)ptr2.back(), 0);               // (one slot per J^2 entry of the engine, also of rows behind n_t)
  out.tcols.reserve((size_t)n_t * 3);
  std::vector<int32_t> stamp((size_t)n_cols, -1), pos((size_t)n_cols, 0), others;
  int cap2 = 1;
  for (int t = 0; t < nt; ++t) {
    const int c0 = out.trow[(size_t)t], c1 = out.trow[(size_t)t + 1];
    const int base = (int)out.tcols.size();
    for (int c = c0; c < c1; ++c) { stamp[(size_t)c] = t; pos[(size_t)c] = c - c0; out.tcols.push_back(c); }
    others.clear();
    for (int q = ptr2[(size_t)c0]; q < ptr2[(size_t)c1]; ++q) { const int k = col2[(size_t)q]; if (stamp[(size_t)k] != t) { stamp[(size_t)k] = t; others.push_back(k); } }
    std::sort(others.begin(), others.end());
    for (size_t u = 0; u < others.size(); ++u) { pos[(size_t)others[u]] = (c1 - c0) + (int)u; out.tcols.push_back(others[u]); }
    const int ncol = (int)out.tcols.size() - base;
    if ((int64_t)ncol * K > 65535) return false;                   // positions travel pre-multiplied by K in 16 bits
    // (stored pre-multiplied by K: the entry's double index into the tile's x image, so the kernel's inner loop has no multiply)
    for (int q = ptr2[(size_t)c0]; q < ptr2[(size_t)c1]; ++q) out.loc2[(size_t)q] = (uint16_t)(pos[(size_t)col2[(size_t)q]] * K);
    out.tptr[(size_t)t + 1] = (int32_t)out.tcols.size();
    out.max_cols = std::max(out.max_cols, ncol);
    cap2 = std::max(cap2, ptr2[(size_t)c1] - ptr2[(size_t)c0]);
  }
  out.cap2 = cap2 + (cap2 & 1);                                    // even: the 16-bit index array keeps what follows 4-byte aligned
  return true;
}

// ---- wave-sliced entry layout of the tiled pass (round 6) ---------------------------------------------------------------------
// The tiled pass gives a row to a lane group of G lanes, so a wave relaxes rpw = 64 / G consecutive rows of the tile together and runs
// as long as its LONGEST row: with the entries in CSR order every lane carries its own loop bounds (exec masks, a compare and ~7 scalar
// instructions per gathered entry -- profiles/r06_pmc_detail_chained_pass_K16.txt).  Sliced layout (ELLPACK per wave): the rows
// [s rpw, (s + 1) rpw) of a tile form slice s, padded to the slice's longest row L_s and stored entry-major -- entry k of row r of the
// slice at  tile base + slice offset + k rpw + r  -- so the k-th gather of a wave reads rpw consecutive entries, every lane of the wave
// makes the same number of trips (a scalar loop), and the addresses advance by a constant.  Padding entries have weight 0 and the
// position of the row's own cell (a valid, finite x row): + 0 x, exact.  The tile-balanced numbering sorts the rows of a tile by J^2 row
// length, so slices are nearly homogeneous (padding: a few per cent of the entries).
struct EllLayout {
  std::vector<int32_t> eptr;            // [ntiles + 1] first entry of every tile in the sliced arrays
  std::vector<int32_t> sl;              // [ntiles][nsl + 1] slice offsets relative to the tile's first entry
  std::vector<int32_t> pos;             // [nnz2] CSR entry -> its index in the sliced arrays (what the numeric kernels store through)
  std::vector<uint16_t> loc;            // [total] position (pre-multiplied by K) of every sliced entry
  int nsl = 0;                          // slices per tile (tile rows capacity / rpw)
  int cap = 0;                          // most sliced entries of a tile, even
  size_t total() const { return loc.size(); }
};
inline bool build_ell(const Tiling& tl, const std::vector<int32_t>& ptr2, int K, int rpw, int tile_rows_cap, EllLayout& out) {
  out = EllLayout();
  if (rpw < 1 || tile_rows_cap % rpw != 0) return false;
  const int nt = tl.ntiles();
  out.nsl = tile_rows_cap / rpw;
  out.eptr.assign((size_t)nt + 1, 0);
  out.sl.assign((size_t)nt * (out.nsl + 1), 0);
  out.pos.assign((size_t)ptr2.back(), -1);
  int cap = 2;
  for (int t = 0; t < nt; ++t) {
    const int c0 = tl.trow[(size_t)t], c1 = tl.trow[(size_t)t + 1];
    if (c1 - c0 > tile_rows_cap) return false;
    const size_t base = out.loc.size();
    int off = 0;
    for (int sidx = 0; sidx < out.nsl; ++sidx) {
      out.sl[(size_t)t * (out.nsl + 1) + sidx] = off;
      const int r0 = c0 + sidx * rpw, r1 = std::min(c1, r0 + rpw);
      int L = 0;
      for (int c = r0; c < r1; ++c) L = std::max(L, ptr2[(size_t)c + 1] - ptr2[(size_t)c]);
      out.loc.resize(base + (size_t)off + (size_t)L * rpw);
      for (int k = 0; k < L; ++k)
        for (int r = 0; r < rpw; ++r) {
          const int c = r0 + r;
          const size_t idx = base + (size_t)off + (size_t)k * rpw + r;
          if (c < r1 && k < ptr2[(size_t)c + 1] - ptr2[(size_t)c]) {
            const size_t q = (size_t)ptr2[(size_t)c] + k;
            out.pos[q] = (int32_t)idx;
            out.loc[idx] = tl.loc2[q];
          } else {
            out.loc[idx] = (uint16_t)((c < r1 ? (c - c0) : 0) * K);     // padding: weight 0, the row's own cell (own rows come first in a tile's list)
          }
        }
      off += L * rpw;
    }
    out.sl[(size_t)t * (out.nsl + 1) + out.nsl] = off;
    if (off & 1) { out.loc.push_back(0); }                          // (tiles start at even entries: the 16-bit array stays 4-byte aligned per tile)
    out.eptr[(size_t)t + 1] = (int32_t)out.loc.size();
    cap = std::max(cap, off + (off & 1));
    if (out.loc.size() > 2000000000u) return false;
  }
  out.cap = cap;
  return true;
}

// per tile: the ptr2 entries of its rows, then the codes of its virtual items (one prefetch stream in the kernel)
inline std::vector<int32_t> tile_meta(int n_t, const std::vector<int32_t>& ptr2, const Tiling& tl) {
  std::vector<int32_t> meta((size_t)n_t + tl.vtab.size());
  for (int t = 0; t < tl.ntiles(); ++t) {
    size_t m = (size_t)tl.trow[(size_t)t] + (size_t)tl.vptr[(size_t)t];
    for (int c = tl.trow[(size_t)t]; c < tl.trow[(size_t)t + 1]; ++c) meta[m++] = ptr2[(size_t)c];
    for (int v = tl.vptr[(size_t)t]; v < tl.vptr[(size_t)t + 1]; ++v) meta[m++] = (int32_t)tl.vtab[(size_t)v];
  }
  return meta;
}

// interior tiles of a partitioned engine: every row they hold and every x row they read is a core row (no exchange touches them)
inline void split_interior(int n_core, const Tiling& tl, std::vector<int32_t>& inner, std::vector<int32_t>& outer) {
  inner.clear(); outer.clear();
  for (int t = 0; t < tl.ntiles(); ++t) {
    bool in = tl.trow[(size_t)t + 1] <= n_core;
    for (int q = tl.tptr[(size_t)t]; q < tl.tptr[(size_t)t + 1] && in; ++q) in = tl.tcols[(size_t)q] < n_core;
    (in ? inner : outer).push_back(t);
  }
}

// ---- chained passes: links, chains, schedule, carry-over codes ----------------------------------------------------------------
// Directed tile links (tile = row / TR, or the row ranges of `trow`): for every ordered pair of distinct tiles that share a face, the
// adjacency entries (edge code = face index << 1 | side, as in ent_edge) through which a cell of src meets a cell of dst, sorted by
// (src, dst), entries of a link in their adjacency order.
struct TileLinks {
  std::vector<int32_t> src, dst, lptr, lent;
  int n() const { return (int)src.size(); }
};
inline void build_links(int n, int TR, int nt, const std::vector<int32_t>& ptr, const std::vector<int32_t>& nb,
                        const std::vector<int32_t>& edge, TileLinks& out, const std::vector<int32_t>* trow = nullptr) {
  // trow (optional): the tiles' row ranges when they are not all TR rows long (windows cut by build_tiling's limits)
  std::vector<int32_t> tile_of_row;
  if (trow) {
    tile_of_row.assign((size_t)n, 0);
    for (int t = 0; t + 1 < (int)trow->size(); ++t)
      for (int c = (*trow)[(size_t)t]; c < (*trow)[(size_t)t + 1] && c < n; ++c) tile_of_row[(size_t)c] = t;
  }
  auto tile = [&](int c) { return trow ? tile_of_row[(size_t)c] : c / TR; };
  struct Ent { int64_t key; int32_t code; };
  std::vector<Ent> ents;
  for (int c = 0; c < n; ++c)
    for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) {
      const int m = nb[(size_t)j];
      if (m < 0 || m >= n || tile(m) == tile(c)) continue;
      ents.push_back({(int64_t)tile(c) * nt + tile(m), edge[(size_t)j]});           // flow leaving c's side of the face
    }
  std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.key < b.key; });
  out = TileLinks();
  out.lptr.assign(1, 0);
  out.lent.resize(ents.size());
  for (size_t i = 0; i < ents.size(); ++i) {
    if (i == 0 || ents[i].key != ents[i - 1].key) {
      if (i > 0) out.lptr.push_back((int32_t)i);
      out.src.push_back((int32_t)(ents[i].key / nt)); out.dst.push_back((int32_t)(ents[i].key % nt));
    }
    out.lent[i] = ents[i].code;
  }
  out.lptr.push_back((int32_t)ents.size());
}

// tile t -> nxt[t]: the destination of its largest outflow, kept when that tile's largest inflow comes from t; ties go to the
// smaller tile id (links are sorted by (src, dst)).  flux[l] = flow leaving link l's source side (k_link_flux).
inline void chains_from_flux(int nt, const TileLinks& lk, const std::vector<float>& flux, std::vector<int32_t>& nxt) {
  std::vector<int32_t> best_dn((size_t)nt, -1), best_up((size_t)nt, -1);
  std::vector<float> w_dn((size_t)nt, 0.f), w_up((size_t)nt, 0.f);
  nxt.assign((size_t)nt, -1);
  for (int l = 0; l < lk.n(); ++l) {
    const float w = flux[(size_t)l];
    if (!(w > 0.f)) continue;
    const int a = lk.src[(size_t)l], b = lk.dst[(size_t)l];
    if (w > w_dn[(size_t)a]) { w_dn[(size_t)a] = w; best_dn[(size_t)a] = b; }
    if (w > w_up[(size_t)b] || (w == w_up[(size_t)b] && a < best_up[(size_t)b])) { w_up[(size_t)b] = w; best_up[(size_t)b] = a; }
  }
  for (int a = 0; a < nt; ++a) { const int b = best_dn[(size_t)a]; if (b >= 0 && best_up[(size_t)b] == a) nxt[(size_t)a] = b; }
}

// chains -> schedule [depth][grid], -1 padded (schedule.py: the same construction)
// subset (optional): schedule only these tiles -- a chain ends where its successor is not one of them
inline void chains_to_schedule(int nt_all, int grid, int SPB, const std::vector<int32_t>& nxt, std::vector<int32_t>& sched, int& depth,
                               const std::vector<int32_t>* subset = nullptr) {
  std::vector<char> has_prev((size_t)nt_all, 0), seen((size_t)nt_all, subset ? 1 : 0);
  if (subset) for (int32_t t : *subset) seen[(size_t)t] = 0;                // (tiles outside the subset count as visited)
  const int nt = subset ? (int)subset->size() : nt_all;
  if (nt == 0 || grid <= 0) { sched.clear(); depth = 0; return; }
  std::vector<char> member(seen.size());
  for (size_t t = 0; t < seen.size(); ++t) member[t] = !seen[t];
  auto next = [&](int t) { const int u = nxt[(size_t)t]; return (u >= 0 && member[(size_t)u]) ? u : -1; };
  for (int t = 0; t < nt_all; ++t) if (member[(size_t)t] && next(t) >= 0) has_prev[(size_t)next(t)] = 1;
  std::vector<std::vector<int32_t>> chains;
  auto walk = [&](int start) {
    if (seen[(size_t)start]) return;
    std::vector<int32_t> ch;
    for (int c = start; c >= 0 && !seen[(size_t)c]; c = next(c)) { seen[(size_t)c] = 1; ch.push_back(c); }
    chains.push_back(std::move(ch));
  };
  for (int t = 0; t < nt_all; ++t) if (!has_prev[(size_t)t]) walk(t);       // heads first,
  for (int t = 0; t < nt_all; ++t) walk(t);                                  // then whatever sits on a cycle
  std::stable_sort(chains.begin(), chains.end(), [](const std::vector<int32_t>& a, const std::vector<int32_t>& b) { return a[0] < b[0]; });
  std::vector<int32_t> seq; seq.reserve((size_t)nt);
  for (const auto& ch : chains) seq.insert(seq.end(), ch.begin(), ch.end());
  // the chains, in the order of their first tile (along the cell curve: an XCD keeps a compact region), are cut into SPB * grid
  // consecutive STREAMS of equal length (+-1); block b = lidx * 8 + xcd walks streams SPB (xcd * grid / 8 + lidx) ...
  // SPB = 1 (column reuse on): a tile's successor takes the rows they share from LDS, so it simply comes next.
  // SPB = 2 (reuse off): the kernel prefetches a tile's x rows from memory one tile ahead, so a chain successor has to come
  // two slots later to read its predecessor's results: two streams INTERLEAVED (A1 B1 A2 B2 ...)
  const int ns = grid * SPB, bpx = grid / HB_N_XCD;
  auto bound = [&](int s_) { return (int)(((int64_t)s_ * nt) / ns); };
  int longest = 0;
  for (int s_ = 0; s_ < ns; ++s_) longest = std::max(longest, bound(s_ + 1) - bound(s_));
  depth = longest * SPB;
  sched.assign((size_t)depth * grid, -1);
  for (int b = 0; b < grid; ++b) {
    const int xcd = b % HB_N_XCD, lidx = b / HB_N_XCD;
    const int s0 = (xcd * bpx + lidx) * SPB;
    int it = 0;
    for (int i = 0; i < longest; ++i)
      for (int q = 0; q < SPB; ++q) {
        const int lo = bound(s0 + q), hi = bound(s0 + q + 1);
        if (lo + i < hi) sched[(size_t)(it++) * grid + b] = seq[(size_t)(lo + i)];
      }
  }
}

// Per-schedule column lists: a column the PREVIOUS tile of the same list holds in LDS is coded -2 - (its position there).
// scols starts as a copy of the tiles' column lists; only the tiles of `sched` are rewritten.  n_cols = rows a column may name.
inline void reuse_codes(int n_cols, const std::vector<int32_t>& tptr, const std::vector<int32_t>& tcols, const std::vector<int32_t>& sched,
                        int grid, int depth, std::vector<int32_t>& scols) {
  std::vector<int32_t> owner((size_t)n_cols, -1), pos((size_t)n_cols, 0);
  for (int b = 0; b < grid; ++b) {
    int prev = -1;
    for (int it = 0; it < depth; ++it) {
      const int tl = sched[(size_t)it * grid + b];
      if (tl < 0) break;
      if (prev >= 0) {
        for (int q = tptr[(size_t)prev]; q < tptr[(size_t)prev + 1]; ++q) { owner[(size_t)tcols[(size_t)q]] = prev; pos[(size_t)tcols[(size_t)q]] = q - tptr[(size_t)prev]; }
        for (int q = tptr[(size_t)tl]; q < tptr[(size_t)tl + 1]; ++q) {
          const int g = tcols[(size_t)q];
          if (owner[(size_t)g] == prev) scols[(size_t)q] = -2 - pos[(size_t)g];
        }
      }
      prev = tl;
    }
  }
}

// A caller's schedule (cwr_set_tile_schedule): every tile exactly once, lists dense from the top.  Returns nullptr when valid.
inline const char* validate_schedule(int ntiles, int n_lists, int depth, const int32_t* sched) {
  std::vector<char> seen((size_t)ntiles, 0);
  int count = 0;
  for (int b = 0; b < n_lists; ++b) {
    bool ended = false;
    for (int it = 0; it < depth; ++it) {
      const int t = sched[(size_t)it * n_lists + b];
      if (t < 0) { ended = true; continue; }
      if (ended || t >= ntiles || seen[(size_t)t]) return "a tile out of range, listed twice, or behind the end of a list";
      seen[(size_t)t] = 1; ++count;
    }
  }
  return count == ntiles ? nullptr : "every tile must appear exactly once";
}


// ---- the one-launch solver of small and mid-size meshes (k_small_jacobi) -----------------------------------------------------
// One 1024-thread workgroup relaxes up to RPT x 1024 rows whose column lives in its LDS; P workgroups ("parts") share the rows
// of one constituent when the mesh is larger than that, each with `depth` layers of halo rows around its own, exchanged every
// `depth` sweeps through global memory (the engine's multi-GPU partition in miniature: the iterates are those of the global
// Jacobi iteration, bit for bit, whatever P is).
//   order     Cuthill-McKee order of the adjacency (breadth-first levels from a pseudo-peripheral cell; no coordinates needed):
//             the q-th nearest-position neighbour of consecutive positions is a run of consecutive positions, so the 32 lanes of a
//             half-wave gather from different LDS banks; part p owns positions [p L, (p + 1) L) of it;
//   layers    halo layer l of a part = the cells at graph distance l from its own; layers 1 .. depth-1 are relaxed redundantly
//             (exact for depth - l sweeps after an exchange), layer `depth` is only read; all of them are refreshed by the exchange;
//   classes   inside a part a stable sort by the class of the real-neighbour count (7-8, 5-6, up to 4): the 64 rows a wave
//             relaxes together gather the same number of neighbours (four without a branch, then two, then two);
//   per row   its real neighbours in ascending Cuthill-McKee position (ghost faces carry no weight in J: no slot): the record index
//             (for the step's weights) and the byte offset in the part's column, two offsets per word.
struct SmallPlan {
  int P = 0, rpt = 0, depth = 0, threads = 0;
  int S = 0, R = 0;                        // padded lengths of the send / receive lists (0 when P == 1)
  std::vector<int32_t> rows;               // [P][rpt * threads]  global row | kind << 28 (1 own, 2 relaxed halo, 3 read-only halo); -1: none
  std::vector<int32_t> recs;               // [P][8][rpt * threads]  record index of the q-th real neighbour (-1: none)
  std::vector<uint32_t> offs;              // [P][4][rpt * threads]  byte offsets of neighbours 2 qq (low half) and 2 qq + 1 (high half)
  std::vector<int32_t> send_pos, send_cnt; // [P][S] local position of the own row published in slot s; [P]
  std::vector<int32_t> recv_src, recv_pos, recv_cnt;   // [P][R] part * S + slot read; local position written; [P]
  std::vector<int32_t> n_local;            // [P] rows of a part, halo included
};
constexpr int SP_DEG = 8;                  // = cwr::SMALL_DEG

inline void cuthill_mckee(int n, const std::vector<int32_t>& ptr, const std::vector<int32_t>& nb, const std::vector<int32_t>& deg,
                          std::vector<int32_t>& order) {
  order.clear(); order.reserve((size_t)n);
  std::vector<char> seen((size_t)n, 0);
  // breadth-first order from `start` over the cells not yet marked; returns the last cell reached (the farthest level)
  auto bfs = [&](int start, std::vector<int32_t>& out, std::vector<char>& mark) -> int {
    const size_t first = out.size();
    out.push_back(start); mark[(size_t)start] = 1;
    std::vector<int32_t> nbs;
    for (size_t head = first; head < out.size(); ++head) {
      const int c = out[head];
      nbs.clear();
      for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) {
        const int v = nb[(size_t)j];
        if (v >= 0 && v < n && !mark[(size_t)v]) { mark[(size_t)v] = 1; nbs.push_back(v); }
      }
      std::sort(nbs.begin(), nbs.end(), [&](int32_t a, int32_t b) { return deg[(size_t)a] != deg[(size_t)b] ? deg[(size_t)a] < deg[(size_t)b] : a < b; });
      out.insert(out.end(), nbs.begin(), nbs.end());
    }
    return out.back();
  };
  for (int c0 = 0; c0 < n; ++c0) {
    if (seen[(size_t)c0]) continue;
    // pseudo-peripheral start of this component: the far end of a search from its first cell, then the far end of that one
    std::vector<char> tmp_mark(seen);
    std::vector<int32_t> tmp;
    int far = bfs(c0, tmp, tmp_mark);
    tmp_mark = seen; tmp.clear();
    far = bfs(far, tmp, tmp_mark);
    bfs(far, order, seen);
  }
}

// parts == 0: the fewest parts (up to max_parts) whose rows fit rpt_max rows per thread, preferring 3 rows per thread to 4.
// Returns false when a row has more than 8 real neighbours or no admissible plan exists.
inline bool build_small_plan(int n, const std::vector<int32_t>& ptr, const std::vector<int32_t>& nb, int threads, int rpt_max, int parts,
                             int depth, int max_parts, SmallPlan& out) {
  if (n <= 0) return false;
  std::vector<int32_t> deg((size_t)n, 0);
  for (int c = 0; c < n; ++c) {
    for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) deg[(size_t)c] += (nb[(size_t)j] >= 0 && nb[(size_t)j] < n);
    if (deg[(size_t)c] > SP_DEG) return false;
  }
  for (size_t j = 0; j < nb.size(); ++j) if (nb[j] >= n) return false;      // (engines with halo cells of a partition do not come here)
  std::vector<int32_t> order, gpos((size_t)n);
  cuthill_mckee(n, ptr, nb, deg, order);
  for (int p = 0; p < n; ++p) gpos[(size_t)order[(size_t)p]] = p;
  auto cls = [&](int32_t c, int kind) { return kind == 3 ? 2 : (deg[(size_t)c] > 6 ? 0 : (deg[(size_t)c] > 4 ? 1 : 2)); };

  auto attempt = [&](int P, int rpt, SmallPlan& pl) -> bool {
    const int cap = rpt * threads;
    if ((size_t)cap * 8 > 65535 + 8) return false;                          // byte offsets travel as 16-bit halves
    const int D = P == 1 ? 0 : depth;
    const int L = (n + P - 1) / P;
    pl = SmallPlan();
    pl.P = P; pl.rpt = rpt; pl.depth = D; pl.threads = threads;
    pl.rows.assign((size_t)P * cap, -1);
    pl.recs.assign((size_t)P * SP_DEG * cap, -1);
    pl.offs.assign((size_t)P * (SP_DEG / 2) * cap, 0u);
    pl.n_local.assign((size_t)P, 0);
    std::vector<std::vector<int32_t>> local((size_t)P), kind((size_t)P);
    std::vector<int32_t> lpos((size_t)n), dist((size_t)n);
    std::vector<std::vector<int32_t>> want((size_t)P);                       // per OWNER: the rows other parts hold as halo (deduplicated below)
    std::vector<std::vector<std::pair<int32_t, int32_t>>> halo_of((size_t)P); // per part: (global row, local position) of its halo rows
    for (int p = 0; p < P; ++p) {
      const int lo = std::min(n, p * L), hi = std::min(n, (p + 1) * L);
      if (lo >= hi) return false;                                           // (an empty part: fewer parts would do)
      std::fill(dist.begin(), dist.end(), -1);
      std::vector<int32_t>& loc = local[(size_t)p];
      std::vector<int32_t>& kd = kind[(size_t)p];
      for (int q = lo; q < hi; ++q) { loc.push_back(order[(size_t)q]); kd.push_back(1); dist[(size_t)order[(size_t)q]] = 0; }
      size_t layer_lo = 0, layer_hi = loc.size();
      for (int l = 1; l <= D; ++l) {
        std::vector<int32_t> next;
        for (size_t i = layer_lo; i < layer_hi; ++i) {
          const int c = loc[i];
          for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) {
            const int v = nb[(size_t)j];
            if (v >= 0 && dist[(size_t)v] < 0) { dist[(size_t)v] = l; next.push_back(v); }
          }
        }
        std::sort(next.begin(), next.end(), [&](int32_t a, int32_t b) { return gpos[(size_t)a] < gpos[(size_t)b]; });
        layer_lo = loc.size();
        for (int32_t v : next) { loc.push_back(v); kd.push_back(l < D ? 2 : 3); }
        layer_hi = loc.size();
      }
      if ((int)loc.size() > cap) return false;
      // class sort (stable) of the part's rows, kinds carried along
      std::vector<int32_t> idx(loc.size());
      for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int32_t)i;
      std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return cls(loc[(size_t)a], kd[(size_t)a]) < cls(loc[(size_t)b], kd[(size_t)b]); });
      std::vector<int32_t> loc2(loc.size()), kd2(loc.size());
      for (size_t i = 0; i < idx.size(); ++i) { loc2[i] = loc[(size_t)idx[i]]; kd2[i] = kd[(size_t)idx[i]]; }
      loc.swap(loc2); kd.swap(kd2);
      pl.n_local[(size_t)p] = (int32_t)loc.size();
      std::fill(lpos.begin(), lpos.end(), -1);
      for (size_t i = 0; i < loc.size(); ++i) lpos[(size_t)loc[i]] = (int32_t)i;
      std::vector<std::array<int32_t, 3>> nbl;                               // (global Cuthill-McKee position, local position, record)
      for (size_t i = 0; i < loc.size(); ++i) {
        const int c = loc[i];
        pl.rows[(size_t)p * cap + i] = c | (kd[i] << 28);
        nbl.clear();
        if (kd[i] != 3)
          for (int j = ptr[(size_t)c]; j < ptr[(size_t)c + 1]; ++j) {
            const int v = nb[(size_t)j];
            if (v < 0) continue;
            if (lpos[(size_t)v] < 0) return false;                          // (cannot happen: a relaxed row's neighbours are within `depth` layers)
            nbl.push_back({gpos[(size_t)v], lpos[(size_t)v], j});
          }
        // the order of a row's sum is a property of the ROW (ascending Cuthill-McKee position of its neighbours), not of the part
        // that relaxes it: a halo copy of a row is then the same bits as its owner's value
        std::sort(nbl.begin(), nbl.end());
        for (int q = 0; q < SP_DEG; ++q) {
          const uint32_t off = (uint32_t)(q < (int)nbl.size() ? nbl[(size_t)q][1] : (int32_t)i) * 8u;   // empty slot: the row itself, weight zero
          if (q < (int)nbl.size()) pl.recs[((size_t)p * SP_DEG + q) * cap + i] = nbl[(size_t)q][2];
          pl.offs[((size_t)p * (SP_DEG / 2) + q / 2) * cap + i] |= (q & 1) ? (off << 16) : off;
        }
        if (kd[i] != 1) { halo_of[(size_t)p].emplace_back(c, (int32_t)i); want[(size_t)std::min(P - 1, gpos[(size_t)c] / L)].push_back(c); }
      }
    }
    if (P == 1) return true;
    // send lists: per owner the rows wanted by anyone, once each, in Cuthill-McKee order; slot = index in that list
    std::vector<std::vector<int32_t>> slot_of((size_t)P);
    size_t S = 1, R = 1;
    for (int o = 0; o < P; ++o) {
      std::vector<int32_t>& w = want[(size_t)o];
      std::sort(w.begin(), w.end(), [&](int32_t a, int32_t b) { return gpos[(size_t)a] < gpos[(size_t)b]; });
      w.erase(std::unique(w.begin(), w.end()), w.end());
      S = std::max(S, w.size());
      R = std::max(R, halo_of[(size_t)o].size());
    }
    pl.S = (int)S; pl.R = (int)R;
    pl.send_pos.assign((size_t)P * S, 0); pl.send_cnt.assign((size_t)P, 0);
    pl.recv_src.assign((size_t)P * R, 0); pl.recv_pos.assign((size_t)P * R, 0); pl.recv_cnt.assign((size_t)P, 0);
    std::vector<int32_t> slot((size_t)n, -1);
    for (int o = 0; o < P; ++o) {
      const std::vector<int32_t>& w = want[(size_t)o];
      // the owner's local position of each wanted row
      const int cap = rpt * threads;
      std::fill(lpos.begin(), lpos.end(), -1);
      for (int i = 0; i < pl.n_local[(size_t)o]; ++i) {
        const int32_t code = pl.rows[(size_t)o * cap + i];
        if ((code >> 28) == 1) lpos[(size_t)(code & 0x0fffffff)] = i;
      }
      for (size_t s = 0; s < w.size(); ++s) {
        if (lpos[(size_t)w[s]] < 0) return false;                            // (cannot happen: the owner holds its own rows)
        pl.send_pos[(size_t)o * S + s] = lpos[(size_t)w[s]];
        slot[(size_t)w[s]] = (int32_t)s;
      }
      pl.send_cnt[(size_t)o] = (int32_t)w.size();
    }
    for (int p = 0; p < P; ++p) {
      std::vector<std::pair<int32_t, int32_t>> ent;                          // (source code, local position), sorted for coalesced reads
      for (const auto& h : halo_of[(size_t)p]) {
        const int o = std::min(P - 1, gpos[(size_t)h.first] / L);
        if (o == p || slot[(size_t)h.first] < 0) return false;
        ent.emplace_back(o * (int32_t)S + slot[(size_t)h.first], h.second);
      }
      std::sort(ent.begin(), ent.end());
      for (size_t r = 0; r < ent.size(); ++r) { pl.recv_src[(size_t)p * R + r] = ent[r].first; pl.recv_pos[(size_t)p * R + r] = ent[r].second; }
      pl.recv_cnt[(size_t)p] = (int32_t)ent.size();
    }
    return true;
  };

  if (parts > 0) {
    for (int rpt = 1; rpt <= rpt_max; ++rpt)
      if (attempt(parts, rpt, out)) return true;
    return false;
  }
  if (n <= rpt_max * threads) {                                               // one workgroup: 1, 2, 3 or 4 rows per thread
    const int rpt = (n + threads - 1) / threads;
    return attempt(1, rpt, out);
  }
  if (depth < 1) return false;
  for (int rpt : {3, 4})
    for (int P = 2; P <= max_parts; ++P)
      if (rpt <= rpt_max && (long long)P * rpt * threads >= n && attempt(P, rpt, out)) return true;
  return false;
}

}  // namespace host
}  // namespace cwr
