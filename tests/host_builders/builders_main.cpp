// builders_main.cpp -- TEST INFRASTRUCTURE: drives the host-side index builders of the engine (csrc/cwr_host_builders.hpp) on the
// CPU, built by tests/test_host_builders.py with
//     g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -D_GLIBCXX_ASSERTIONS
// so that every out-of-range index, overflow or use of an empty vector in them aborts here instead of becoming an
// out-of-range LDS / global access on the GPU.  Input and output are flat little-endian int32 files of named arrays:
//     [count of arrays] then per array: [name length][name bytes][element count][elements ...]
// (floats travel as their bit patterns).  The pytest compares the outputs with numpy statements of the same constructions.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "../../clearwater-riverine_amd/csrc/cwr_host_builders.hpp"

using namespace cwr::host;
typedef std::map<std::string, std::vector<int32_t>> Bag;

static Bag read_bag(const char* path) {
  Bag bag;
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  auto rd = [&]() { int32_t v = 0; if (std::fread(&v, 4, 1, f) != 1) { std::fprintf(stderr, "short read\n"); std::exit(2); } return v; };
  const int n = rd();
  for (int i = 0; i < n; ++i) {
    const int ln = rd();
    std::string name((size_t)ln, ' ');
    if (ln > 0 && std::fread(&name[0], 1, (size_t)ln, f) != (size_t)ln) std::exit(2);
    const int cnt = rd();
    std::vector<int32_t> v((size_t)cnt);
    if (cnt > 0 && std::fread(v.data(), 4, (size_t)cnt, f) != (size_t)cnt) std::exit(2);
    bag[name] = std::move(v);
  }
  std::fclose(f);
  return bag;
}
static void write_bag(const char* path, const Bag& bag) {
  FILE* f = std::fopen(path, "wb");
  if (!f) { std::perror(path); std::exit(2); }
  auto wr = [&](int32_t v) { std::fwrite(&v, 4, 1, f); };
  wr((int32_t)bag.size());
  for (const auto& kv : bag) {
    wr((int32_t)kv.first.size());
    std::fwrite(kv.first.data(), 1, kv.first.size(), f);
    wr((int32_t)kv.second.size());
    if (!kv.second.empty()) std::fwrite(kv.second.data(), 4, kv.second.size(), f);
  }
  std::fclose(f);
}
template <typename T> static std::vector<int32_t> widen(const std::vector<T>& v) { return std::vector<int32_t>(v.begin(), v.end()); }

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: builders_main <in> <out>\n"); return 2; }
  Bag in = read_bag(argv[1]), out;
  if (in.count("small_params")) {                      // the plan of the one-launch solver: n, threads, rpt_max, parts, depth, max_parts
    const std::vector<int32_t>& sp = in.at("small_params");
    SmallPlan pl;
    const bool ok = build_small_plan(sp.at(0), in.at("ptr"), in.at("nb"), sp.at(1), sp.at(2), sp.at(3), sp.at(4), sp.at(5), pl);
    out["ok"] = {ok ? 1 : 0};
    if (ok) {
      out["dims"] = {pl.P, pl.rpt, pl.depth, pl.threads, pl.S, pl.R};
      out["rows"] = pl.rows; out["recs"] = pl.recs; out["offs"] = std::vector<int32_t>(pl.offs.begin(), pl.offs.end());
      out["send_pos"] = pl.send_pos; out["send_cnt"] = pl.send_cnt; out["recv_src"] = pl.recv_src; out["recv_pos"] = pl.recv_pos;
      out["recv_cnt"] = pl.recv_cnt; out["n_local"] = pl.n_local;
    }
    write_bag(argv[2], out);
    return 0;
  }
  const std::vector<int32_t>& par = in.at("params");   // n_owned, n_core, n_real, K, tr, grid, seg, nvmax
  const int n_owned = par.at(0), n_core = par.at(1), n_real = par.at(2), K = par.at(3), tr = par.at(4), grid = par.at(5), seg = par.at(6), nvmax = par.at(7);
  const std::vector<int32_t>&ptr = in.at("ptr"), &nb = in.at("nb"), &edge = in.at("edge"), &adv_bits = in.at("adv");
  int max_degree = 0;
  for (int c = 0; c < n_owned; ++c) max_degree = std::max(max_degree, ptr.at((size_t)c + 1) - ptr.at((size_t)c));

  SqPattern sq;
  const bool ok = symbolic_sq(n_owned, n_core, max_degree, ptr, nb, sq);
  out["sq_ok"] = {ok ? 1 : 0};
  if (!ok) { write_bag(argv[2], out); return 0; }
  out["n_sq"] = {sq.n_sq, sq.max_row, sq.rowwise ? 1 : 0};
  out["ptr2"] = sq.ptr2; out["col2"] = sq.col2; out["pair_ptr"] = sq.pair_ptr; out["slots"] = widen(sq.slots); out["fast"] = widen(sq.fast);

  Tiling tl;
  // params[8], [9] (optional): limits on the distinct x rows / J^2 entries of a tile -- heavier windows are cut (round 5)
  const int col_limit = par.size() > 8 ? par.at(8) : (1 << 30), ent_limit = par.size() > 9 ? par.at(9) : (1 << 30);
  int heavy = 0;
  const bool tiled = build_tiling(sq.n_sq, tr, seg, nvmax, K, n_real, sq.ptr2, sq.col2, tl, col_limit, ent_limit, &heavy);
  out["heavy"] = {heavy};
  out["tiled"] = {tiled ? 1 : 0};
  if (!tiled) { write_bag(argv[2], out); return 0; }
  out["trow"] = tl.trow; out["vptr"] = tl.vptr; out["vtab"] = widen(tl.vtab); out["tptr"] = tl.tptr; out["tcols"] = tl.tcols;
  out["loc2"] = widen(tl.loc2); out["tile_dims"] = {tl.max_cols, tl.cap2, tl.ntiles()};
  out["meta"] = tile_meta(sq.n_sq, sq.ptr2, tl);
  std::vector<int32_t> inner, outer;
  split_interior(n_core, tl, inner, outer);
  out["inner"] = inner; out["outer"] = outer;
  // the wave-sliced entry layout of the tiled pass (round 6): params[10] = rows per wave, params[11] = row capacity of a tile (0: skip)
  if (par.size() > 11 && par.at(10) > 0 && heavy == 0 && nvmax == 0) {
    EllLayout ell;
    const bool eok = build_ell(tl, sq.ptr2, K, par.at(10), par.at(11), ell);
    out["ell_ok"] = {eok ? 1 : 0};
    if (eok) {
      out["ell_eptr"] = ell.eptr; out["ell_sl"] = ell.sl; out["ell_pos"] = ell.pos; out["ell_loc"] = widen(ell.loc);
      out["ell_dims"] = {ell.nsl, ell.cap, (int32_t)ell.total()};
    }
  }

  if (seg >= (1 << 20)) {                              // fixed-size tiles: links, chains, schedules, carry-over codes
    const int nt = tl.ntiles();
    TileLinks lk;
    build_links(sq.n_sq, tr, nt, ptr, nb, edge, lk, heavy > 0 ? &tl.trow : nullptr);
    out["link_src"] = lk.src; out["link_dst"] = lk.dst; out["link_ptr"] = lk.lptr; out["link_ent"] = lk.lent;
    // k_link_flux on the CPU: the flow leaving the source side, summed in entry order, stored as float
    std::vector<float> flux((size_t)lk.n());
    std::vector<int32_t> flux_bits((size_t)lk.n());
    for (int l = 0; l < lk.n(); ++l) {
      double s = 0.0;
      for (int j = lk.lptr.at((size_t)l); j < lk.lptr.at((size_t)l + 1); ++j) {
        const int code = lk.lent.at((size_t)j);
        float a; std::memcpy(&a, &adv_bits.at((size_t)(code >> 1)), 4);
        s += std::fmax((code & 1) ? -(double)a : (double)a, 0.0);
      }
      flux[(size_t)l] = (float)s;
      std::memcpy(&flux_bits[(size_t)l], &flux[(size_t)l], 4);
    }
    out["link_flux"] = flux_bits;
    std::vector<int32_t> nxt;
    chains_from_flux(nt, lk, flux, nxt);
    out["nxt"] = nxt;
    for (int spb = 1; spb <= 2; ++spb) {
      std::vector<int32_t> sched; int depth = 0;
      chains_to_schedule(nt, grid, spb, nxt, sched, depth);
      out[spb == 1 ? "sched1" : "sched2"] = sched;
      out[spb == 1 ? "depth1" : "depth2"] = {depth};
      if (const char* why = validate_schedule(nt, grid, depth, sched.data())) { std::fprintf(stderr, "own schedule invalid: %s\n", why); return 3; }
      if (spb == 1) {
        std::vector<int32_t> scols(tl.tcols);
        reuse_codes(n_real, tl.tptr, tl.tcols, sched, grid, depth, scols);
        out["scols"] = scols;
      }
    }
    // the interior / cut split of a partitioned engine: two sub-schedules over one shared copy of the codes
    if (!inner.empty() && !outer.empty()) {
      const int gi = std::max(HB_N_XCD, std::min(grid, ((int)inner.size() + HB_N_XCD - 1) / HB_N_XCD * HB_N_XCD));
      const int go = std::max(HB_N_XCD, std::min(grid, ((int)outer.size() + HB_N_XCD - 1) / HB_N_XCD * HB_N_XCD));
      std::vector<int32_t> s_in, s_out; int d_in = 0, d_out = 0;
      chains_to_schedule(nt, gi, 1, nxt, s_in, d_in, &inner);
      chains_to_schedule(nt, go, 1, nxt, s_out, d_out, &outer);
      std::vector<int32_t> scols(tl.tcols);
      reuse_codes(n_real, tl.tptr, tl.tcols, s_in, gi, d_in, scols);
      reuse_codes(n_real, tl.tptr, tl.tcols, s_out, go, d_out, scols);
      out["sched_in"] = s_in; out["sched_out"] = s_out; out["sub_dims"] = {gi, d_in, go, d_out}; out["scols_io"] = scols;
    }
  }
  write_bag(argv[2], out);
  return 0;
}
