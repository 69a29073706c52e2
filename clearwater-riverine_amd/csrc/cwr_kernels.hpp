// cwr_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the transport engine.
//
// Reference semantics implemented here (paths relative to /root/reference/src/clearwater_riverine/):
//   k_derive_coeff   utilities.py:513-535      advection_coeff / edge_vertical_area / coeff_to_diffusion
//   k_prep_step      linalg.py:34-156          the per-step operator coefficients, regrouped per cell
//   k_rhs            linalg.py:177-201,227-275,354-406   right-hand side incl. ghost (boundary) terms
//   k_apply          transport.py:215-218 (A = csr(coo)) applied matrix-free: y = A x
//   k_vec_s/k_vec_x  the BiCGSTAB recurrences standing in for transport.py:249 (spsolve)
//   k_ghost_writeback transport.py:258-264, constituents.py:39-48
//   k_mass_flux      transport.py:406-429
//   k_sq_tiled       two fused Jacobi sweeps per tile-local application, x tile in LDS: the dominant kernel; walks the default
//                    tile order (ping-pong passes) or a per-block list of tiles chained along the flow, in place, carrying the
//                    columns consecutive tiles share over in LDS (round 3)
//   k_jnorm, k_link_flux   ||J||_inf of every level (scale of the element-wise stopping rule); flow between tiles (chains)
//
// Data layout (all device resident):
//   x[cell*K + k]  float64, constituents inner.  A thread owns VW (1 or 2) consecutive constituents
//   of one cell; G = K / VW consecutive lanes own one cell row, so a wave touches whole 128-B rows
//   at K = 16 and every global access of a vector is fully coalesced.
//   FaceRec rec[j]  one 16-byte record per (cell, face) adjacency entry, entries of a cell contiguous
//   (CSR over ptr[]), faces of a cell in ascending face id: {neighbour cell or -1-ghost, signed
//   outflow a_c (float32, as HEC-RAS stores it), diffusion coefficient d (float64)}.
//   No MFMA anywhere: this is a bandwidth-bound sparse gather, ~0.1 flop/byte.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace cwr {

constexpr int BLOCK = 256;          // 4 waves of 64
constexpr int N_XCD = 8;            // MI355X: 8 XCDs, workgroups dealt round-robin over them
constexpr int ACC_R0V = 0, ACC_TS = 1, ACC_TT = 2, ACC_R0T = 3, ACC_R0S = 4, ACC_RR = 5, ACC_N = 6;

struct __attribute__((aligned(16))) FaceRec {
  int32_t nb;      // >= 0: local real cell id (owned or halo); < 0: ghost cell -1-nb (boundary, not in A)
  float a_c;       // flow leaving THIS cell through the face (advection_coeff with the cell's sign)
  double d;        // coeff_to_diffusion of the face
};

// Scalars of the batched BiCGSTAB (K independent systems sharing A), all on device so that a whole
// batch of iterations can be enqueued without a host round trip.  acc is a ring of 3 slots:
// iteration `it` writes its inner products into slot it%3 and reads ||r||^2 of iteration it-1 from
// slot (it-1)%3, so no kernel ever reads a row another kernel of the same iteration rewrites.
struct SolverScalars {
  double* acc;       // [3][ACC_N][K]
  double* rho;       // [3][K]
  double* bb;        // [K]   ||D^-1 b||^2
  int32_t* counters; // [0] iterations with an active column, [1] breakdown flag, [2] ghost-coefficient
                     // precondition violated, [3] non-finite seen
};

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  // Workgroups are dealt round-robin over the 8 XCDs (observed, speed only): give each XCD one
  // contiguous range of cell blocks so a face's two visits and a cell row's ~5 gathers hit one L2.
  const int per = (nblocks + N_XCD - 1) / N_XCD;
  return (bid % N_XCD) * per + bid / N_XCD;
}

template <int VW> struct VecT;
template <> struct VecT<1> { using type = double; };
template <> struct VecT<2> { using type = double2; };

template <int VW> __device__ __forceinline__ void ldv(const double* p, double (&v)[VW]) {
  if constexpr (VW == 4) {
    const double2 t = *reinterpret_cast<const double2*>(p), u = *reinterpret_cast<const double2*>(p + 2);
    v[0] = t.x; v[1] = t.y; v[2] = u.x; v[3] = u.y;
  } else
  if constexpr (VW == 2) { const double2 t = *reinterpret_cast<const double2*>(p); v[0] = t.x; v[1] = t.y; }
  else { v[0] = *p; }
}
// streamed-once operands (face records, bhat / r0, outputs) bypass cache retention with `nt` loads/stores so the
// gathered x rows keep the L2 (measured on MI355X: -6 % per step at K = 16, +12 % at K = 1, so the host enables it per engine for wide rows only)
#ifndef CWR_NT
#define CWR_NT 1
#endif
template <int VW> __device__ __forceinline__ void ldv_nt(const double* p, double (&v)[VW]) {
#if CWR_NT
  if constexpr (VW == 4) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 t = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p)), u = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + 2));
    v[0] = t.x; v[1] = t.y; v[2] = u.x; v[3] = u.y;
  } else if constexpr (VW == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 t = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p)); v[0] = t.x; v[1] = t.y;
  } else { v[0] = __builtin_nontemporal_load(p); }
#else
  ldv<VW>(p, v);
#endif
}
// CWR_STREAM_STORE: 0 plain, 1 nt, 2 sc1 (write-through, line not kept in this XCD's L2)
#ifndef CWR_STREAM_STORE
#define CWR_STREAM_STORE 0
#endif
template <int VW> __device__ __forceinline__ void stv_stream(double* p, const double (&v)[VW]) {
#if CWR_STREAM_STORE == 2
  if constexpr (VW == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 t; t.x = v[0]; t.y = v[1];
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
  } else {
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v[0]) : "memory");
  }
#elif CWR_STREAM_STORE == 1
  if constexpr (VW == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 t; t.x = v[0]; t.y = v[1];
    __builtin_nontemporal_store(t, reinterpret_cast<d2*>(p));
  } else { __builtin_nontemporal_store(v[0], p); }
#else
  if constexpr (VW == 4) { *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); *reinterpret_cast<double2*>(p + 2) = make_double2(v[2], v[3]); }
  else if constexpr (VW == 2) { *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); } else { *p = v[0]; }
#endif
}
template <int VW> __device__ __forceinline__ void stv_nt(double* p, const double (&v)[VW]) {
#if CWR_NT
  if constexpr (VW == 2) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 t; t.x = v[0]; t.y = v[1];
    __builtin_nontemporal_store(t, reinterpret_cast<d2*>(p));
  } else { __builtin_nontemporal_store(v[0], p); }
#else
  if constexpr (VW == 2) { *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); } else { *p = v[0]; }
#endif
}
template <int VW> __device__ __forceinline__ void stv(double* p, const double (&v)[VW]) {
  if constexpr (VW == 4) { *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); *reinterpret_cast<double2*>(p + 2) = make_double2(v[2], v[3]); }
  else if constexpr (VW == 2) { *reinterpret_cast<double2*>(p) = make_double2(v[0], v[1]); }
  else { *p = v[0]; }
}

// Column-wise block reduction of NV = ND*VW per-thread partial sums.  The block's totals are STORED to
// out[d*K + g*VW + w] (this block's private slot of a partials buffer): no atomics, so inner products
// are bitwise reproducible; k_reduce_partials folds the slots in a fixed order afterwards.  Threads
// are laid out tid = r*G + g (row-in-tile, lane group); when G divides 64 the rows of a wave are
// folded with xor-shuffles (DPP / ds_bpermute class cross-lane ops), otherwise through LDS.
// Dots with index >= MAXFROM are folded with max instead of + (the element-wise convergence measures of MODE 4).
template <int NV, int VW, int MAXFROM = 1 << 20>
__device__ __forceinline__ void block_reduce_cols(double (&val)[NV], int G, int K, double* out, double* lds) {
  const int tid = threadIdx.x;
  constexpr int ND = NV / VW;
  if ((64 % G) == 0) {
    for (int off = 32; off >= G; off >>= 1) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const double o = __shfl_xor(val[i], off, 64);
        val[i] = (i / VW >= MAXFROM) ? fmax(val[i], o) : val[i] + o;
      }
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < G) {
#pragma unroll
      for (int i = 0; i < NV; ++i) lds[(wave * G + lane) * NV + i] = val[i];
    }
    __syncthreads();
    if (tid < G) {
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int w = 0; w < VW; ++w) {
          const int i = d * VW + w;
          double s = lds[tid * NV + i];
          for (int wv = 1; wv < BLOCK / 64; ++wv) { const double o = lds[(wv * G + tid) * NV + i]; s = (d >= MAXFROM) ? fmax(s, o) : s + o; }
          out[d * K + tid * VW + w] = s;
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[tid * NV + i] = val[i];
    __syncthreads();
    if (tid < G) {
      const int R = BLOCK / G;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int w = 0; w < VW; ++w) {
          const int i = d * VW + w;
          double s = lds[tid * NV + i];
          for (int r = 1; r < R; ++r) { const double o = lds[(r * G + tid) * NV + i]; s = (d >= MAXFROM) ? fmax(s, o) : s + o; }
          out[d * K + tid * VW + w] = s;
        }
    }
  }
}

// Second stage of every inner product: out[d][k] = sum over slots of partial[slot][d][k], slots folded in a
// fixed order (strided per thread, then an LDS tree), one block per dot d.  `outs` holds the destination
// row of each dot (they are not contiguous for the start-up residual: ||r||^2 and ||bhat||^2).
struct ReduceOuts { double* p[4]; };
// (round 5) The convergence check without a copy and without draining the stream: the folded scalars are ALSO stored into a page-locked
// host buffer the GPU can write (host_out[d * K + k]); the last of the ND blocks to finish publishes a sequence number behind them
// (system-scope release), and the host spins on that word.  The sequence lives in device memory (dev_seq) because these launches
// are captured into hipGraphs: a by-value argument would freeze it.  All null: no notification.
struct ReduceNote { double* host_out; unsigned long long* host_seq; unsigned int* arrive; unsigned long long* dev_seq; };
// Partitioned engines: the check scalars of this rank laid out for the ONE all-reduce (sum) of a check -- [rr | bb] to be added,
// [m1 | m2] in the rank's own slot of a (world x 2K) block that is zero elsewhere, the zero-coefficient flag last -- in one launch
// (round 5; a memset and three device-to-device copies before).
__global__ void __launch_bounds__(256) k_pack_check(int n, int K, int rank, const double* __restrict__ chk, const double* __restrict__ flag,
                                                    double* __restrict__ out) {
  const int lo = 2 * K + rank * 2 * K;
  for (int i = threadIdx.x; i < n; i += 256) {
    double v = 0.0;
    if (i < 2 * K) v = chk[i];
    else if (i >= lo && i < lo + 2 * K) v = chk[2 * K + (i - lo)];
    else if (i == n - 1) v = *flag;
    out[i] = v;
  }
}
// ... and the all-reduced block written to page-locked host memory with a sequence word behind it: the host spins on the word
// instead of enqueueing a copy and draining the stream (as k_reduce_partials does for a single engine).
__global__ void __launch_bounds__(256) k_note_out(int n, const double* __restrict__ src, double* __restrict__ host_out,
                                                  unsigned long long* __restrict__ host_seq, unsigned long long* __restrict__ dev_seq) {
  for (int i = threadIdx.x; i < n; i += 256) host_out[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long sq = *dev_seq + 1ull;
    *dev_seq = sq;
    __hip_atomic_store(host_seq, sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
constexpr int RBLOCK = 1024;
__global__ void __launch_bounds__(RBLOCK) k_reduce_partials(int nslots, int ND, int K, const double* __restrict__ partial,
                                                          ReduceOuts outs, int max_from, ReduceNote note) {
  __shared__ double sm[RBLOCK];
  const int d = blockIdx.x;
  double* out = outs.p[d];
  if (out == nullptr) return;                   // uniform per block (never with a notification: the host passes every dot then)
  const int S = RBLOCK / K;                     // slot lanes per column
  const int tid = threadIdx.x;
  const int k = tid % K, l = tid / K;
  const size_t stride = (size_t)ND * K;
  const double* base = partial + (size_t)d * K + k;
  if (d >= max_from) {                          // dots folded with max (uniform per block); -inf is the identity
    double m = -INFINITY;
    if (l < S) for (int slot = l; slot < nslots; slot += S) m = fmax(m, base[(size_t)slot * stride]);
    sm[tid] = m;
    __syncthreads();
    for (int len = S; len > 1;) {                // fixed tree over the slot lanes (see below)
      const int h = (len + 1) >> 1;
      if (l < len - h) sm[tid] = fmax(sm[tid], sm[tid + h * K]);
      __syncthreads();
      len = h;
    }
  } else {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // four independent chains keep four loads in flight
    if (l < S) {
      int slot = l;
      for (; slot + 3 * S < nslots; slot += 4 * S) {
        s0 += base[(size_t)slot * stride];
        s1 += base[(size_t)(slot + S) * stride];
        s2 += base[(size_t)(slot + 2 * S) * stride];
        s3 += base[(size_t)(slot + 3 * S) * stride];
      }
      for (; slot < nslots; slot += S) s0 += base[(size_t)slot * stride];
    }
    sm[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    // the slot lanes of a column are folded in a fixed tree (lane l takes lane l + ceil(len / 2)): log2(S) steps instead of one
    // thread adding S values one after the other -- at K = 1 that was 1 024 dependent additions, 10 of the launch's 16 us
    for (int len = S; len > 1;) {
      const int h = (len + 1) >> 1;
      if (l < len - h) sm[tid] += sm[tid + h * K];
      __syncthreads();
      len = h;
    }
  }
  if (tid < K) out[tid] = sm[tid];
  if (note.host_seq) {
    if (tid < K) note.host_out[(size_t)d * K + tid] = sm[tid];
    __threadfence_system();                     // this block's values are out before it says so
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = atomicAdd(note.arrive, 1u);
      if (prev == (unsigned)ND - 1u) {          // the last block: every other block's values are visible (fence / atomic / fence)
        __threadfence_system();
        *note.arrive = 0u;                      // (the next notifying launch is stream-ordered behind this one)
        const unsigned long long sq = *note.dev_seq + 1ull;
        *note.dev_seq = sq;
        __hip_atomic_store(note.host_seq, sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// Internal face order: the engine stores every per-face array sorted by the smaller cell id of the face, so that the
// faces of neighbouring cells are neighbours in memory (per-entry gathers of k_prep_step, both cell rows of k_mass_flux).
// dst[t, p] = src[t, orig[p]] on the way in, dst[orig[p]] = src[p] on the way out; ids at the C ABI stay the reference's.
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_faces_in(int64_t total, int E, const int32_t* __restrict__ orig,
                                                  const T* __restrict__ src, T* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t t = i / E; const int p = (int)(i - t * E);
    dst[i] = src[t * E + orig[p]];
  }
}
// rows of K values: dst[orig[p], :] = src[p, :] (per-face K-vectors from the internal face order to the reference's)
// (src2: a second array added element by element -- total_mass_flux = advection + diffusion is formed on the way out, see k_mass_flux)
__global__ void __launch_bounds__(BLOCK) k_face_rows_out(int64_t total, int K, int Kp, const int32_t* __restrict__ orig,
                                                       const double* __restrict__ src, const double* __restrict__ src2,
                                                       double* __restrict__ dst) {
  // K: the caller's constituents (the rows of dst), Kp >= K: the engine's internal row width (zero columns behind K: cwr_create)
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t p = i / K; const int k = (int)(i - p * K);
    const size_t j = (size_t)p * Kp + k;
    dst[(size_t)orig[p] * K + k] = src2 ? src[j] + src2[j] : src[j];
  }
}
// rows of K values <-> rows of Kp >= K values (the engine carries zero columns behind the caller's K where that is faster)
__global__ void __launch_bounds__(BLOCK) k_pad_cols(int64_t total, int K, int Kp, const double* __restrict__ src, double* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t r = i / Kp; const int k = (int)(i - r * Kp);
    dst[i] = k < K ? src[r * K + k] : 0.0;
  }
}
__global__ void __launch_bounds__(BLOCK) k_strip_cols(int64_t total, int K, int Kp, const double* __restrict__ src, double* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t r = i / K; const int k = (int)(i - r * K);
    dst[i] = src[r * Kp + k];
  }
}
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_faces_out(int E, const int32_t* __restrict__ orig, const T* __restrict__ src,
                                                   T* __restrict__ dst) {
  const int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p < E) dst[orig[p]] = src[p];
}

// ------------------------------------------------------------------------------------------------ a-1
// utilities.py:514-535, one thread per (time level, face).
__global__ void __launch_bounds__(BLOCK) k_derive_coeff(
    int64_t total, int E, const float* __restrict__ flow, const float* __restrict__ vel,
    const double* __restrict__ dist, float Df, float* __restrict__ adv, double* __restrict__ dif) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int e = (int)(i % E);
    const float f = flow[i], v = vel[i];
    // np.sign(abs(v)): NaN -> NaN, 0 -> 0, else 1
    const float sg = (v != v) ? v : (fabsf(v) > 0.0f ? 1.0f : 0.0f);
    const float a = f * sg;
    float area = a / v;                     // float32 division, correctly rounded (hipcc default)
    if (area != area) area = 0.0f;          // .fillna(0)
    const float ad = area * Df;             // float32 array * python float stays float32
    adv[i] = a;
    dif[i] = (double)ad / dist[e];
  }
}

// ------------------------------------------------------------------------------------------------ a-2
// Regroup the coefficients of level t per (cell, face) entry and form the diagonal
//   diag[c] = V[t+1,c]/dt + [V[t+1,c]==0] + sum_faces ( d + max(a_c, 0) )
// (linalg.py:76-103 dry dummy, V/dt, diffusion diagonals; :113-115 outflow incl. ghost faces;
//  :139-141 inflow seen from the neighbour).  One thread per owned cell.
constexpr int PREP_CAP = 1536;      // adjacency entries of a 256-row block staged in LDS (4-6 per row on HEC-RAS meshes)
__device__ __forceinline__ void prep_block_rows(
    int n_owned, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_edge,
    const int32_t* __restrict__ ent_nb, const float* __restrict__ adv_t, const double* __restrict__ dif_t,
    const float* __restrict__ vol_next, double dt, FaceRec* __restrict__ rec, double* __restrict__ diag,
    double* __restrict__ w) {
  // The block's adjacency entries are gathered ENTRY-parallel (entry i by thread i mod 256: the code and neighbour loads
  // are coalesced, the two face gathers of all of a thread's entries are in flight together), staged in LDS, reduced per
  // row by the row's thread, and leave as one contiguous stream (a thread's own 16-byte stores at a 64-byte stride cost
  // 2.6 x the bytes at the HBM side: PMC, profiles/r01_k_*).  One thread per row walking its own entries -- round 1's
  // form -- made every load wait for the one before it (65.8 -> 58.1 us per level, profiles/r02_u_set_up_kernels.txt).
  __shared__ FaceRec s_rec[PREP_CAP];
  __shared__ double s_w[PREP_CAP];
  const int c0 = blockIdx.x * BLOCK, c1 = min(c0 + BLOCK, n_owned);
  const int jb = ptr[c0], nent = ptr[c1] - jb;
  const int c = c0 + threadIdx.x;
  if (nent <= PREP_CAP) {                            // uniform per block
    constexpr int NU = PREP_CAP / BLOCK;
    int code[NU], nb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = threadIdx.x + u * BLOCK;
      code[u] = 0; nb[u] = -1;
      if (i < nent) { code[u] = ent_edge[jb + i]; nb[u] = ent_nb[jb + i]; }
    }
    float a[NU]; double d[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      a[u] = 0.f; d[u] = 0.0;
      if (threadIdx.x + u * BLOCK < nent) { a[u] = adv_t[code[u] >> 1]; d[u] = dif_t[code[u] >> 1]; }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = threadIdx.x + u * BLOCK;
      if (i < nent) { FaceRec r; r.nb = nb[u]; r.a_c = (code[u] & 1) ? -a[u] : a[u]; r.d = d[u]; s_rec[i] = r; }
    }
    __syncthreads();
    if (c < n_owned) {
      const double vn = (double)vol_next[c];
      double dg = vn / dt + (vn == 0.0 ? 1.0 : 0.0);
      const int j0 = ptr[c] - jb, j1 = ptr[c + 1] - jb;
      for (int j = j0; j < j1; ++j) dg += s_rec[j].d + fmax((double)s_rec[j].a_c, 0.0);   // (ascending j: the reference's order)
      diag[c] = dg;
      // w[j] = -offd_j / diag >= 0: the Jacobi iteration matrix J = I - D^-1 A per adjacency entry (0 on ghost faces)
      for (int j = j0; j < j1; ++j) {
        const FaceRec fr = s_rec[j];
        s_w[j] = (fr.nb >= 0) ? (fr.d - fmin((double)fr.a_c, 0.0)) / dg : 0.0;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nent; i += BLOCK) { rec[jb + i] = s_rec[i]; w[jb + i] = s_w[i]; }
  } else if (c < n_owned) {                          // denser adjacency than the stage holds: one thread per row, straight to memory
    const double vn = (double)vol_next[c];
    double dg = vn / dt + (vn == 0.0 ? 1.0 : 0.0);
    const int j0 = ptr[c], j1 = ptr[c + 1];
    for (int j = j0; j < j1; ++j) {
      const int code = ent_edge[j];
      const int e = code >> 1;
      const float a = adv_t[e];
      const double d = dif_t[e];
      const float a_c = (code & 1) ? -a : a;
      dg += d + fmax((double)a_c, 0.0);
      FaceRec r; r.nb = ent_nb[j]; r.a_c = a_c; r.d = d;
      rec[j] = r;
    }
    diag[c] = dg;
    for (int j = j0; j < j1; ++j) {
      const FaceRec fr = rec[j];
      w[j] = (fr.nb >= 0) ? (fr.d - fmin((double)fr.a_c, 0.0)) / dg : 0.0;
    }
  }
}

__global__ void __launch_bounds__(BLOCK) k_prep_step(
    int n_owned, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_edge,
    const int32_t* __restrict__ ent_nb, const float* __restrict__ adv_t, const double* __restrict__ dif_t,
    const float* __restrict__ vol_next, double dt, FaceRec* __restrict__ rec, double* __restrict__ diag,
    double* __restrict__ w) {
  prep_block_rows(n_owned, ptr, ent_edge, ent_nb, adv_t, dif_t, vol_next, dt, rec, diag, w);
}

// The reference's zero-coefficient ValueError (linalg.py:349-351) depends on the flow field only: flags[t] = 1 when
// level t has an active ghost face (edge_velocity != 0) of a computed row whose advection coefficient (inflow) or
// diffusion coefficient (either direction, D != 0) is exactly 0 -- the condition k_rhs tests at level t+1 of a step.
// Evaluated once per loaded flow field, so a single-GPU step needs no device round trip to raise the error.
__global__ void __launch_bounds__(BLOCK) k_check_ghost_levels(
    int64_t total, int E, int n_owned, int n_real, const int32_t* __restrict__ f1, const int32_t* __restrict__ f2,
    const float* __restrict__ vel, const float* __restrict__ adv, const double* __restrict__ dif, int use_diffusion,
    int32_t* __restrict__ flags) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
    const int64_t t = i / E; const int e = (int)(i - t * E);
    if (f2[e] < n_real || f1[e] >= n_owned) continue;
    const float v = vel[i];
    if (!(v < 0.0f) && !(v > 0.0f)) continue;
    const bool d0 = use_diffusion && fabs(dif[i]) == 0.0;
    if (v < 0.0f ? (fabs((double)adv[i]) == 0.0 || d0) : d0) flags[t] = 1;
  }
}

// ||J||_inf of every loaded level: jn[t] = max over the computed rows of sum_j w[j] = (sum over faces with a real neighbour of
// (d - min(a_c, 0))) / diag, the coefficients k_prep_step forms for step t (a, d of level t; V of level t+1).  The exact
// max-norm contraction of the Jacobi iteration x <- J x + bhat, so ||x* - x'||_inf <= jn / (1 - jn) ||x' - x||_inf: what the
// element-wise stopping rule is scaled by (cwr_step).  A property of the flow field: evaluated once when it is loaded.
// grid (row blocks, T - 1); non-negative doubles order like their bit patterns, so the fold is an integer atomicMax.
__global__ void __launch_bounds__(BLOCK) k_jnorm(
    int n_owned, int E, int n_cells, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_edge,
    const int32_t* __restrict__ ent_nb, const float* __restrict__ adv, const double* __restrict__ dif,
    const float* __restrict__ vol_next, const double* __restrict__ dt, unsigned long long* __restrict__ jn, double dt_one) {
  // adv / dif: level t of the launch's first step at offset 0 (stride E); vol_next: V of level t + 1 at offset 0 (stride n_cells)
  // grid (row blocks, levels): the rows are walked grid-stride, so the host bounds the blocks per level -- one same-address atomic per
  // block is what the launch waits for (3 906 blocks per level at 1 M cells: 39 us per level; <= 1 024 now)
  __shared__ double s_m[BLOCK / 64];
  const int t = blockIdx.y;
  double rho = 0.0;
  for (int c = blockIdx.x * BLOCK + threadIdx.x; c < n_owned; c += gridDim.x * BLOCK) {
    const float* adv_t = adv + (size_t)t * E;
    const double* dif_t = dif + (size_t)t * E;
    const double vn = (double)vol_next[(size_t)t * n_cells + c];
    double dg = vn / (dt ? dt[t] : dt_one) + (vn == 0.0 ? 1.0 : 0.0), off = 0.0;     // (dt == nullptr: one step, its dt by value)
    const int j0 = ptr[c], j1 = ptr[c + 1];
    // the first eight faces of the row with their loads batched (codes, then coefficients: two rounds of independent loads instead of
    // a dependent pair per face -- the kernel was latency-bound: 38 us per level at 1 M cells), summed in the same order as before
    {
      int code[8], nbv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool live = j0 + q < j1;
        code[q] = live ? ent_edge[j0 + q] : -1;
        nbv[q] = live ? ent_nb[j0 + q] : -1;
      }
      float a8[8]; double d8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int f = code[q] >= 0 ? (code[q] >> 1) : 0;
        a8[q] = adv_t[f]; d8[q] = dif_t[f];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (code[q] >= 0) {
          const double a_c = (code[q] & 1) ? -(double)a8[q] : (double)a8[q];
          dg += d8[q] + fmax(a_c, 0.0);
          if (nbv[q] >= 0) off += d8[q] - fmin(a_c, 0.0);
        }
      }
    }
    for (int j = j0 + 8; j < j1; ++j) {
      const int code = ent_edge[j];
      const float a = adv_t[code >> 1];
      const double d = dif_t[code >> 1];
      const double a_c = (code & 1) ? -(double)a : (double)a;
      dg += d + fmax(a_c, 0.0);
      if (ent_nb[j] >= 0) off += d - fmin(a_c, 0.0);
    }
    double rr = off / dg;
    if (!(rr >= 0.0)) rr = INFINITY;               // NaN in the field: no bound
    rho = fmax(rho, rr);
  }
  for (int o = 32; o >= 1; o >>= 1) rho = fmax(rho, __shfl_xor(rho, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = rho;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = s_m[0];
    for (int w = 1; w < BLOCK / 64; ++w) m = fmax(m, s_m[w]);
    atomicMax(jn + t, (unsigned long long)__double_as_longlong(m));
  }
}

// One level of a windowed flow field on its way in (cwr_flow_window_load), in ONE pass over the faces: the face flows and velocities
// arrive in the reference's face order (in_flow, in_vel) and leave in the internal one -- k_faces_in twice --, the coefficients are
// derived (k_derive_coeff: utilities.py:514-535, the same float32 / float64 steps) and the level's zero-coefficient flag is set
// (k_check_ghost_levels: linalg.py:349-351).  The first thread zeroes the flag and the norm word of the step this level completes
// first (everything later on the stream sets / folds into them).
__global__ void __launch_bounds__(BLOCK) k_level_in(
    int E, int n_owned, int n_real, const int32_t* __restrict__ orig, const int32_t* __restrict__ f1, const int32_t* __restrict__ f2,
    const float* __restrict__ in_flow, const float* __restrict__ in_vel, const double* __restrict__ dist, float Df, int use_diffusion,
    float* __restrict__ vel, float* __restrict__ adv, double* __restrict__ dif, int32_t* __restrict__ flag) {
  for (int p = blockIdx.x * BLOCK + threadIdx.x; p < E; p += gridDim.x * BLOCK) {
    const int src = orig[p];
    const float f = in_flow[src], v = in_vel[src];
    const float sg = (v != v) ? v : (fabsf(v) > 0.0f ? 1.0f : 0.0f);
    const float a = f * sg;
    float area = a / v;
    if (area != area) area = 0.0f;
    const float ad = area * Df;
    const double d = (double)ad / dist[p];
    vel[p] = v; adv[p] = a; dif[p] = d;
    if (f2[p] >= n_real && f1[p] < n_owned && ((v < 0.0f) || (v > 0.0f))) {
      const bool d0 = use_diffusion && fabs(d) == 0.0;
      if (v < 0.0f ? (fabs((double)a) == 0.0 || d0) : d0) flag[0] = 1;
    }
  }
}

// windowed flow fields: the per-level scalars the host needs at the step, left in page-locked memory by the flow stream
// (the zero-coefficient flag of a level as 0.0 / 1.0; ||J||_inf of up to two steps from their bit patterns)
__global__ void k_note_level(const int32_t* __restrict__ flag, double* __restrict__ flag_out, const unsigned long long* __restrict__ jn_a,
                             double* __restrict__ out_a, const unsigned long long* __restrict__ jn_b, double* __restrict__ out_b) {
  flag_out[0] = flag[0] ? 1.0 : 0.0;
  if (jn_a) out_a[0] = __longlong_as_double((long long)jn_a[0]);
  if (jn_b) out_b[0] = __longlong_as_double((long long)jn_b[0]);
  __threadfence_system();
}

// (round 6) ... of a PARTITIONED engine: a rank's values of an arriving level -- the zero-coefficient flag, ||J||_inf over its own rows of up
// to two steps -- laid out for ONE sum all-reduce (every rank in a slot of its own of a world x 3 block that is zero elsewhere, as
// gather_check lays out the convergence check), and behind the all-reduce the fold over the ranks (any flag; the largest norm; NaN from
// any rank stays NaN) into the same page-locked words a single engine's k_note_level writes.
__global__ void __launch_bounds__(64) k_pack_level(int world, int rank, const int32_t* __restrict__ flag, const unsigned long long* __restrict__ jn_a,
                                                   const unsigned long long* __restrict__ jn_b, double* __restrict__ block) {
  for (int i = threadIdx.x; i < 3 * world; i += 64) block[i] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    block[3 * rank + 0] = flag[0] ? 1.0 : 0.0;
    block[3 * rank + 1] = jn_a ? __longlong_as_double((long long)jn_a[0]) : 0.0;
    block[3 * rank + 2] = jn_b ? __longlong_as_double((long long)jn_b[0]) : 0.0;
  }
}
__global__ void __launch_bounds__(64) k_note_level_ranks(int world, const double* __restrict__ block, double* __restrict__ flag_out,
                                                         double* __restrict__ out_a, double* __restrict__ out_b) {
  if (threadIdx.x != 0) return;
  double f = 0.0, a = 0.0, b = 0.0;
  for (int r = 0; r < world; ++r) {
    const double fr = block[3 * r], ar = block[3 * r + 1], br = block[3 * r + 2];
    if (fr != 0.0) f = 1.0;
    a = (ar != ar || a != a) ? NAN : fmax(a, ar);
    b = (br != br || b != b) ? NAN : fmax(b, br);
  }
  flag_out[0] = f;
  if (out_a) out_a[0] = a;
  if (out_b) out_b[0] = b;
  __threadfence_system();
}

__global__ void __launch_bounds__(BLOCK) k_fill(int64_t total, double v, double* __restrict__ a, double* __restrict__ b) {
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) { a[i] = v; if (b) b[i] = v; }
}

// One sweep w' = 1 + J w of the Neumann series of (I - J)^-1 1 (the row-wise a-posteriori factor of the element-wise rule: see
// refine_error_factors in cwr_engine.hip), MATRIX-FREE and on ONE column: J's entries are formed on the fly from the level's
// coefficients exactly as k_prep_step forms them -- no operator buffers, no K-wide vectors -- so the sweeps of an incoming level can
// run on the flow-field stream beside the steps of a windowed run.  One thread per computed row; rows >= n_owned of w (a rank's
// read-only layer) are input only.  max[0] = max over rows < n_dot of |w' - w| (from w = 1 the series is monotone: w' - w >= 0), max[1] = max w',
// folded with integer atomicMax on the bit patterns (non-negative doubles order like their bits; the host zeroes the two words).
__global__ void __launch_bounds__(BLOCK) k_neumann(
    int n_owned, int n_dot, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_edge, const int32_t* __restrict__ ent_nb,
    const float* __restrict__ adv_t, const double* __restrict__ dif_t, const float* __restrict__ vol_next, double dt,
    const double* __restrict__ win, double* __restrict__ wout, unsigned long long* __restrict__ maxima) {
  // maxima == nullptr: a sweep nobody checks (all but the last of a batch) -- no reduction at all.  Grid-stride over the rows with a
  // bounded grid: the two same-address atomics per block are what a checked sweep waits for (3 906 blocks at 1 M cells: 111 us per
  // sweep, profiles/r05_final_kernel_stats_bench_K16.csv of the first build; <= 1 024 blocks and only the last sweep of a batch now).
  __shared__ double s_r[BLOCK / 64], s_w[BLOCK / 64];
  double r = 0.0, wv = 0.0;
  for (int c = blockIdx.x * BLOCK + threadIdx.x; c < n_owned; c += gridDim.x * BLOCK) {
    const double vn = (double)vol_next[c];
    double dg = vn / dt + (vn == 0.0 ? 1.0 : 0.0), sum = 0.0;
    const int j1 = ptr[c + 1];
    for (int j = ptr[c]; j < j1; ++j) {
      const int code = ent_edge[j];
      const float a = adv_t[code >> 1];
      const double d = dif_t[code >> 1];
      const double a_c = (code & 1) ? -(double)a : (double)a;
      dg += d + fmax(a_c, 0.0);
      const int nb = ent_nb[j];
      if (nb >= 0) sum += (d - fmin(a_c, 0.0)) * win[nb];
    }
    double w1 = 1.0 + sum / dg;
    wout[c] = w1;
    if (maxima && c < n_dot) {
      double dr = fabs(w1 - win[c]);                            // (|r|: a warm start's iterates are not monotone -- refine_level)
      if (dr != dr) dr = INFINITY;                              // (NaN in the field: no bound)
      if (!(w1 >= 0.0)) w1 = INFINITY;
      r = fmax(r, dr); wv = fmax(wv, w1);
    }
  }
  if (!maxima) return;                                          // (uniform)
  for (int o = 32; o >= 1; o >>= 1) { r = fmax(r, __shfl_xor(r, o, 64)); wv = fmax(wv, __shfl_xor(wv, o, 64)); }
  if ((threadIdx.x & 63) == 0) { s_r[threadIdx.x >> 6] = r; s_w[threadIdx.x >> 6] = wv; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double mr = s_r[0], mw = s_w[0];
    for (int q = 1; q < BLOCK / 64; ++q) { mr = fmax(mr, s_r[q]); mw = fmax(mw, s_w[q]); }
    atomicMax(maxima, (unsigned long long)__double_as_longlong(mr));
    atomicMax(maxima + 1, (unsigned long long)__double_as_longlong(mw));
  }
}

// ------------------------------------------------------------------------------------------------ a-3
// b[c,k] = V[t,c]*x[c,k]/dt + G_in[c,k] + G_out[c,k]; boundary terms from level t+1, selected by the
// sign of edge_velocity[t+1]; the highest active ghost-face id of a cell wins in each set
// (linalg.py:349-351,378: assignment, not accumulation).  SCALE: divide by diag (Jacobi row scaling).
// rows c = c_first, c_first + c_step, ... < c_end of the right-hand side (one lane group per row; col = the lane's first constituent)
template <int VW, bool SCALE>
__device__ __forceinline__ void rhs_rows(
    int c_first, int c_end, int c_step, int K, int col, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_edge,
    const int32_t* __restrict__ ent_nb, const float* __restrict__ vol_t, double dt,
    const float* __restrict__ vel_n, const float* __restrict__ adv_n, const double* __restrict__ dif_n,
    int use_diffusion, const double* __restrict__ bc_n, const double* __restrict__ x,
    const double* __restrict__ diag, const uint8_t* __restrict__ row_ghost, double* __restrict__ b,
    int32_t* __restrict__ counters, double* __restrict__ x_keep, double* __restrict__ bad_flag) {
  for (int c = c_first; c < c_end; c += c_step) {
  double xv[VW], gin[VW], gout[VW];
  bool bad = false;
  ldv<VW>(x + (size_t)c * K + col, xv);
  // x_t is kept aside: the solver iterates in place in the state vector, and a step that fails (no convergence, NaN)
  // must leave the state as it found it so that the caller can retry (cwr_step restores it from this copy)
  if (x_keep) stv<VW>(x_keep + (size_t)c * K + col, xv);
#pragma unroll
  for (int w = 0; w < VW; ++w) { gin[w] = 0.0; gout[w] = 0.0; }
  // only rows with a boundary (ghost) face walk their entries: for the others the two dependent loads (row pointer ->
  // neighbour id) would be pure latency
  const int j1 = row_ghost[c] ? ptr[c + 1] : 0;
  for (int j = row_ghost[c] ? ptr[c] : 0; j < j1; ++j) {
    const int nb = ent_nb[j];
    if (nb >= 0) continue;
    const int e = ent_edge[j] >> 1;
    const float v1 = vel_n[e];
    if (!(v1 < 0.0f) && !(v1 > 0.0f)) continue;
    const double a = fabs((double)adv_n[e]);
    const double d = use_diffusion ? fabs(dif_n[e]) : 0.0;
    double cg[VW];
    ldv<VW>(bc_n + (size_t)(-1 - nb) * K + col, cg);
    if (v1 < 0.0f) {
      if (a == 0.0 || (use_diffusion && d == 0.0)) { counters[2] = 1; bad_flag[0] = 1.0; bad = true; }
#pragma unroll
      for (int w = 0; w < VW; ++w) gin[w] = (a + d) * cg[w];
    } else {
      if (use_diffusion && d == 0.0) { counters[2] = 1; bad_flag[0] = 1.0; bad = true; }
#pragma unroll
      for (int w = 0; w < VW; ++w) gout[w] = d * cg[w];
    }
  }
  const double vt = (double)vol_t[c];
  double out[VW];
#pragma unroll
  for (int w = 0; w < VW; ++w) {
    out[w] = (vt * xv[w] / dt + gin[w]) + gout[w];
    if (SCALE) out[w] /= diag[c];
    // a violated precondition poisons the row: the residual turns NaN on EVERY rank after the all-reduce, so a
    // partitioned run stops everywhere instead of leaving the other ranks waiting in a collective
    if (bad) out[w] = __builtin_nan("");
  }
  stv<VW>(b + (size_t)c * K + col, out);
  }
}

template <int VW, bool SCALE>
__global__ void __launch_bounds__(BLOCK) k_rhs(
    int n_owned, int K, int G, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_edge,
    const int32_t* __restrict__ ent_nb, const float* __restrict__ vol_t, double dt,
    const float* __restrict__ vel_n, const float* __restrict__ adv_n, const double* __restrict__ dif_n,
    int use_diffusion, const double* __restrict__ bc_n, const double* __restrict__ x,
    const double* __restrict__ diag, const uint8_t* __restrict__ row_ghost, double* __restrict__ b,
    int32_t* __restrict__ counters, double* __restrict__ x_keep, int keep_from, int keep_rows, double* __restrict__ ew_out, double ew_val,
    double* __restrict__ bad_flag) {
  const int R = BLOCK / G;
  const int r = threadIdx.x / G, g = threadIdx.x - r * G;
  // this step's relative element-wise tolerance, for every MODE 4 sweep that follows (k_apply reads it from memory)
  if (ew_out && blockIdx.x == 0 && threadIdx.x == 0) ew_out[0] = ew_val;
  if (r >= R) return;
  const int col = g * VW;
  rhs_rows<VW, SCALE>(blockIdx.x * R + r, n_owned, gridDim.x * R, K, col, ptr, ent_edge, ent_nb, vol_t, dt, vel_n, adv_n, dif_n, use_diffusion,
                      bc_n, x, diag, row_ghost, b, counters, x_keep, bad_flag);      // grid-stride: a block works many row groups
  // the ghost rows of the state as the step found them (rows keep_from ...): kept with x_t for a failed step
  if (x_keep) {
    for (int c = keep_from + blockIdx.x * R + r; c < keep_from + keep_rows; c += gridDim.x * R) {
      double v[VW];
      ldv<VW>(x + (size_t)c * K + col, v);
      stv<VW>(x_keep + (size_t)c * K + col, v);
    }
  }
}

// (round 5) The opening of a step in ONE launch: the operator of level t for the block's 256 rows (k_prep_step), their scaled
// right-hand sides with x_t kept aside (k_rhs), the kept copy of the rows behind the computed ones (halo, ghost) -- and the ghost
// rows' values of level t+1 (k_ghost_writeback: transport.py:258-264), which no kernel of the solve reads and the step's closing
// flux kernel needs; a failed step puts the kept rows back.  Three launches and a drained stream between each pair less per step:
// what counts on engines of ~100 k cells, whose set-up kernels are all latency (profiles/r05_*).
template <int VW>
__global__ void __launch_bounds__(BLOCK) k_begin_step(
    int n_owned, int n_real, int n_cells, int K, int G, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_edge,
    const int32_t* __restrict__ ent_nb, const float* __restrict__ adv_t, const double* __restrict__ dif_t,
    const float* __restrict__ vol_next, double dt, FaceRec* __restrict__ rec, double* __restrict__ diag, double* __restrict__ w,
    const float* __restrict__ vol_t, const float* __restrict__ vel_n, const float* __restrict__ adv_n, const double* __restrict__ dif_n,
    int use_diffusion, const double* __restrict__ bc_n, double* __restrict__ x, const uint8_t* __restrict__ row_ghost,
    double* __restrict__ b, int32_t* __restrict__ counters, double* __restrict__ x_keep, double* __restrict__ ew_out, double ew_val,
    double* __restrict__ bad_flag) {
  if (blockIdx.x == 0 && threadIdx.x == 0) ew_out[0] = ew_val;
  prep_block_rows(n_owned, ptr, ent_edge, ent_nb, adv_t, dif_t, vol_next, dt, rec, diag, w);
  __syncthreads();                                   // diag of the block's rows is in memory (block scope)
  const int R = BLOCK / G;
  const int r = threadIdx.x / G, g = threadIdx.x - r * G;
  if (r >= R) return;
  const int col = g * VW;
  const int c0 = blockIdx.x * BLOCK, c1 = min(c0 + BLOCK, n_owned);
  rhs_rows<VW, true>(c0 + r, c1, R, K, col, ptr, ent_edge, ent_nb, vol_t, dt, vel_n, adv_n, dif_n, use_diffusion, bc_n, x, diag, row_ghost, b,
                     counters, x_keep, bad_flag);
  for (int c = n_owned + blockIdx.x * R + r; c < n_cells; c += gridDim.x * R) {
    double v[VW];
    ldv<VW>(x + (size_t)c * K + col, v);
    stv<VW>(x_keep + (size_t)c * K + col, v);
    if (c >= n_real) {                               // ghost row: boundary value of level t+1 where non-zero, NaN otherwise
      double nv[VW];
      ldv<VW>(bc_n + (size_t)(c - n_real) * K + col, nv);
#pragma unroll
      for (int q = 0; q < VW; ++q) nv[q] = (nv[q] != 0.0) ? nv[q] : __builtin_nan("");
      stv<VW>(x + (size_t)c * K + col, nv);
    }
  }
}

// ------------------------------------------------------------------------------------------------ the operator
// y[c,:] = diag[c]*x[c,:] + sum over the faces of c with a real neighbour of ( -d + min(a_c,0) ) * x[nb,:]
// MODE 0  y = A x                                   (cwr_apply; parity against the oracle's csr @ x)
// MODE 1  v = D^-1 A p ;            partial[R0V] = (r0, v)                     (BiCGSTAB, first product)
// MODE 2  t = D^-1 A s ;            partial[TS,TT,R0T,R0S] = (t,s), (t,t), (r0,t), (r0,s)   (second product)
// MODE 3  r = bhat - D^-1 A x ; r0 = p = r ; partial = (r,r), (bhat,bhat)      (start / verify)
// MODE 4  x' = x + (bhat - D^-1 A x) = bhat - (sum offd x[nb]) / diag ;  partial = (x'-x, x'-x), (bhat,bhat)
//         one fully fused Jacobi sweep: ||x'-x|| IS the scaled residual of x, so convergence needs no
//         second pass, no recurrence and (between checks) no reduction at all
// MODE 5  x'' = c2 + J^2 x  with J = I - D^-1 A, c2 = bhat + J bhat: TWO Jacobi sweeps in one launch.  ptr/rec are
//         the CSR of J^2 (built per step by k_build_sq; rec.d carries the coefficient, rec.nb the column), so the
//         pass moves x, c2 and x'' once for two iterations; its gathers (13 rows per row on a quad mesh) are L2 hits
//
// Persistent grid, XCD-aware: the row tiles (TR = U*R rows each) are split into 8 contiguous ranges, one
// per XCD (blockIdx % 8 names the XCD group under round-robin dispatch -- speed only), and the blocks of
// an XCD sweep their range together, so the ~5 gathers of a cell row and the two visits of a face meet
// in that XCD's L2.  Per tile the face records are staged through LDS with one coalesced 16-B load per
// lane; each lane group then walks its rows' records from LDS (broadcast reads) and gathers the
// neighbour rows.  Inner-product partials stay in registers across tiles and leave the block once.
// Dynamic LDS: [stage_cap FaceRec][reduction scratch][TR + 1 row pointers].
__host__ __device__ inline int red_doubles(int G, int VW) {     // scratch of block_reduce_cols for <= 4 dots
  return ((64 % G) == 0 ? (BLOCK / 64) * G : BLOCK) * 4 * VW;
}

#ifndef CWR_FACE_BATCH
#define CWR_FACE_BATCH 4          // faces whose neighbour gathers are in flight together (tuning knob)
#endif
constexpr int TCL_SEG = 12;         // J^2 entries per work item of the tiled pass (see k_sq_tiled); rows are summed in chunks of it
constexpr int TCL_NVMAX = 48;       // virtual items (chunks 1.. of long rows) per tile
#ifndef CWR_APPLY_MIN_WAVES
#define CWR_APPLY_MIN_WAVES 1     // __launch_bounds__ second argument: waves per SIMD the allocator must allow
#endif
template <int VW, int MODE>
__global__ void __launch_bounds__(BLOCK, CWR_APPLY_MIN_WAVES) k_apply(
    int row0, int n_owned, int n_dot, int K, int G, int U, int ntiles, int stage_cap, int nt, const int32_t* __restrict__ ptr,
    const FaceRec* __restrict__ rec, const double* __restrict__ diag, const double* __restrict__ xin,
    double* __restrict__ yout, const double* __restrict__ r0, const double* __restrict__ bhat,
    double* __restrict__ r0_out, double* __restrict__ p_out, double* __restrict__ partial, const double* __restrict__ ew_p, int seg,
    const int32_t* __restrict__ tile_list, int slot0) {
  // tile_list (optional): the launch covers these `ntiles` row tiles only (the closing sweep of a partitioned engine in two parts:
  // core tiles beside the halo exchange, cut tiles behind it); slot0: first slot of this launch in the partials buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  // ew_rel lives in device memory (written by k_rhs at the start of every step): these launches are captured into hipGraphs
  // that later steps replay, and a by-value argument would freeze the value of the step that captured them
  const double ew_rel = (MODE == 4) ? ew_p[0] : 0.0;
  FaceRec* s_rec = reinterpret_cast<FaceRec*>(s_dyn);
  double* s_red = reinterpret_cast<double*>(s_dyn + (size_t)stage_cap * sizeof(FaceRec));
  int32_t* s_ptr = reinterpret_cast<int32_t*>(s_red + red_doubles(G, VW));
  const int R = BLOCK / G;
  const int TR = R * U;
  const int tid = threadIdx.x;
  const int r = tid / G, g = tid - r * G;
  const int col = g * VW;
  // MODE 4 carries two more partials, folded with max: the element-wise convergence measures of the sweep
  //   [2] max_i ( |x'_i - x_i| - ew_rel |x'_i| )      [3] max_i |x'_i|
  // (host: converged element-wise when [2] <= ew_abs * [3]; see solve_jacobi)
  constexpr int ND = (MODE == 1) ? 1 : ((MODE == 2 || MODE == 4) ? 4 : ((MODE == 3 || MODE == 5) ? 2 : 0));
  constexpr int NP = (ND > 0 ? ND : 1) * VW;
  constexpr int MAXFROM = (MODE == 4) ? 2 : (1 << 20);
  double part[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) part[i] = (MODE == 4 && i >= 2 * VW && i < 3 * VW) ? -INFINITY : 0.0;

  const int xcd = blockIdx.x % N_XCD, lidx = blockIdx.x / N_XCD, bpx = gridDim.x / N_XCD;
  const int tpx = (ntiles + N_XCD - 1) / N_XCD;
  for (int i = lidx; i < tpx; i += bpx) {
    const int tslot = xcd * tpx + i;
    if (tslot >= ntiles) break;                     // uniform per block
    const int tile = tile_list ? tile_list[tslot] : tslot;
    const int c0 = row0 + tile * TR;                // rows [row0, n_owned) are this launch's
    const int c1 = min(c0 + TR, n_owned);
    __syncthreads();                                // previous tile's readers are done with the LDS images
    for (int q = tid; q <= c1 - c0; q += BLOCK) s_ptr[q] = ptr[c0 + q];
    __syncthreads();
    const int jb = s_ptr[0], je = s_ptr[c1 - c0];      // the host sizes stage_cap to the largest tile: always fits
    {
      typedef int i4 __attribute__((ext_vector_type(4)));
      if (nt) {
        for (int j = jb + tid; j < je; j += BLOCK)
          reinterpret_cast<i4*>(s_rec)[j - jb] = __builtin_nontemporal_load(reinterpret_cast<const i4*>(rec) + j);
      } else {
        for (int j = jb + tid; j < je; j += BLOCK) s_rec[j - jb] = rec[j];
      }
    }
    __syncthreads();
    if (r < R) {
      const char* xbase = reinterpret_cast<const char*>(xin);
      const unsigned rowB = (unsigned)K * 8u, colB = (unsigned)col * 8u;   // byte offsets stay below 4 GiB (checked at create)
      for (int c = c0 + r; c < c1; c += R) {
        const size_t o = (size_t)c * K + col;
        const double wdot = (c < n_dot) ? 1.0 : 0.0;   // rows of the inner halo layers are computed but belong to a neighbour
        double xc[VW], sum[VW], q0[VW];
        ldv<VW>(xin + o, xc);
        double dg = 1.0;
        if constexpr (MODE != 5) dg = diag[c];
        if constexpr (MODE == 1 || MODE == 2) { if (nt) ldv_nt<VW>(r0 + o, q0); else ldv<VW>(r0 + o, q0); }
        if constexpr (MODE == 3 || MODE == 4 || MODE == 5) { if (nt) ldv_nt<VW>(bhat + o, q0); else ldv<VW>(bhat + o, q0); }
#pragma unroll
        for (int w = 0; w < VW; ++w) sum[w] = 0.0;
        const int j0 = s_ptr[c - c0] - jb, j1 = s_ptr[c - c0 + 1] - jb;
        constexpr int FB = CWR_FACE_BATCH;
        double tot[VW];                               // MODE 5: closed chunks of `seg` entries (the tiled pass's association)
#pragma unroll
        for (int w = 0; w < VW; ++w) tot[w] = 0.0;
        int until = seg;                              // (seg is a multiple of FB: a chunk closes at a batch boundary)
        for (int j = j0; j < j1; j += FB) {
          if constexpr (MODE == 5) {
            if (until == 0) {
#pragma unroll
              for (int w = 0; w < VW; ++w) { tot[w] += sum[w]; sum[w] = 0.0; }
              until = seg;
            }
            until -= FB;
          }
          // up to FB faces at a time, branch-free: a slot past the row's end or a ghost face re-reads the row's own x
          // with a zero coefficient; every neighbour-row gather is issued before the first is consumed
          double coef[FB];
          double xn[FB][VW];
#pragma unroll
          for (int u = 0; u < FB; ++u) {
            const FaceRec fr = s_rec[min(j + u, j1 - 1)];
            const bool ok = (j + u < j1) & (fr.nb >= 0);
            const double cf = (MODE == 5) ? fr.d : fmin((double)fr.a_c, 0.0) - fr.d;
            coef[u] = ok ? cf : 0.0;
            const unsigned row = ok ? (unsigned)fr.nb : (unsigned)c;
            ldv<VW>(reinterpret_cast<const double*>(xbase + (row * rowB + colB)), xn[u]);
          }
#pragma unroll
          for (int u = 0; u < FB; ++u) {
#pragma unroll
            for (int w = 0; w < VW; ++w) sum[w] += coef[u] * xn[u][w];
          }
        }
        if constexpr (MODE == 5) {
#pragma unroll
          for (int w = 0; w < VW; ++w) sum[w] = tot[w] + sum[w];
        }
        double y[VW];
        if constexpr (MODE == 0) {
#pragma unroll
          for (int w = 0; w < VW; ++w) y[w] = dg * xc[w] + sum[w];
          stv<VW>(yout + o, y);
        } else if constexpr (MODE == 4 || MODE == 5) {
#pragma unroll
          for (int w = 0; w < VW; ++w) {
            y[w] = (MODE == 5) ? q0[w] + sum[w] : q0[w] - sum[w] / dg;
            const double dx = y[w] - xc[w];
            part[0 * VW + w] += wdot * dx * dx;
            part[1 * VW + w] += wdot * q0[w] * q0[w];
            if constexpr (MODE == 4) {
              if (c < n_dot) {
                part[2 * VW + w] = fmax(part[2 * VW + w], fabs(dx) - ew_rel * fabs(y[w]));
                part[3 * VW + w] = fmax(part[3 * VW + w], fabs(y[w]));
              }
            }
          }
          stv_stream<VW>(yout + o, y);
        } else {
#pragma unroll
          for (int w = 0; w < VW; ++w) y[w] = xc[w] + sum[w] / dg;
          if constexpr (MODE == 1) {
            stv<VW>(yout + o, y);
#pragma unroll
            for (int w = 0; w < VW; ++w) part[w] += wdot * q0[w] * y[w];
          } else if constexpr (MODE == 2) {
            stv<VW>(yout + o, y);
#pragma unroll
            for (int w = 0; w < VW; ++w) {
              part[0 * VW + w] += wdot * y[w] * xc[w];
              part[1 * VW + w] += wdot * y[w] * y[w];
              part[2 * VW + w] += wdot * q0[w] * y[w];
              part[3 * VW + w] += wdot * q0[w] * xc[w];
            }
          } else {  // MODE 3
            double res[VW];
#pragma unroll
            for (int w = 0; w < VW; ++w) res[w] = q0[w] - y[w];
            stv<VW>(yout + o, res);
            stv<VW>(r0_out + o, res);
            stv<VW>(p_out + o, res);
#pragma unroll
            for (int w = 0; w < VW; ++w) { part[0 * VW + w] += wdot * res[w] * res[w]; part[1 * VW + w] += wdot * q0[w] * q0[w]; }
          }
        }
      }
    }
  }
  if constexpr (ND > 0) {
    __syncthreads();
    block_reduce_cols<NP, VW, MAXFROM>(part, G, K, partial + (size_t)(slot0 + blockIdx.x) * ND * K, s_red);
  }
}

// A/B variant of the operator (north star: "ds_bpermute vs global atomicAdd chosen by rocprof"):
// face-parallel scatter with float64 global atomics.  k_scatter_diag writes y = diag*x, then
// k_scatter_faces adds both sides' off-diagonal terms of every internal face.
template <int VW>
__global__ void __launch_bounds__(BLOCK) k_scatter_diag(int n_owned, int K, int G, const double* __restrict__ diag,
                                                      const double* __restrict__ x, double* __restrict__ y) {
  const int R = BLOCK / G;
  const int r = threadIdx.x / G, g = threadIdx.x - r * G;
  const int c = blockIdx.x * R + r;
  if (r >= R || c >= n_owned) return;
  double xv[VW]; ldv<VW>(x + (size_t)c * K + g * VW, xv);
  const double dg = diag[c];
#pragma unroll
  for (int w = 0; w < VW; ++w) xv[w] *= dg;
  stv<VW>(y + (size_t)c * K + g * VW, xv);
}
template <int VW>
__global__ void __launch_bounds__(BLOCK) k_scatter_faces(int E, int n_owned, int n_real, int K, int G,
                                                       const int32_t* __restrict__ f1, const int32_t* __restrict__ f2,
                                                       const float* __restrict__ adv_t, const double* __restrict__ dif_t,
                                                       const double* __restrict__ x, double* __restrict__ y) {
  const int R = BLOCK / G;
  const int r = threadIdx.x / G, g = threadIdx.x - r * G;
  const int e = blockIdx.x * R + r;
  if (r >= R || e >= E) return;
  const int P = f1[e], N = f2[e];
  if (N >= n_real) return;                          // ghost face: diagonal only (already in diag)
  const double a = (double)adv_t[e], d = dif_t[e];
  double xp[VW], xn[VW];
  ldv<VW>(x + (size_t)P * K + g * VW, xp);
  ldv<VW>(x + (size_t)N * K + g * VW, xn);
  const double offP = fmin(a, 0.0) - d;             // row P, column N
  const double offN = fmin(-a, 0.0) - d;            // row N, column P
#pragma unroll
  for (int w = 0; w < VW; ++w) {
    if (P < n_owned) atomicAdd(&y[(size_t)P * K + g * VW + w], offP * xn[w]);
    if (N < n_owned) atomicAdd(&y[(size_t)N * K + g * VW + w], offN * xp[w]);
  }
}

// ------------------------------------------------------------------------------------------------ squared operator
// (the Jacobi weights w[j] = -offd_j / diag[row] >= 0 of J = I - D^-1 A are written by k_prep_step)
// Numeric J^2, row-wise: one thread per row c.  The host lists, for every product J[c,m] J[m,k] in the order (faces of
// c ascending, then faces of m ascending; ghost faces skipped), the slot of column k in row c of J^2; the thread
// accumulates into its own LDS row (strided: conflict-free) and writes the row out.  Same summation order as
// k_build_sq (which stays for rows longer than SQN_MAXC): bitwise the same values.
constexpr int SQN_THREADS = 128;
constexpr int SQN_PAD = 8;          // elements the weight and slot arrays are over-allocated by (whole-row vector loads of k_sq_numeric)
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));
typedef int32_t int2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t uint32_u __attribute__((aligned(1)));
constexpr int SQN_MAXC = 255;           // longest J^2 row the row-wise kernel takes (slots are 8-bit)
// The accumulators of a block's 128 rows sit in LDS in the rows' own CSR layout (entry q of row c at ptr2[c] - ptr2[c0] + q;
// the host sizes the dynamic LDS to the fullest block), so a block holds ~10 KB and 32 waves per CU hide the dependent
// L2 latencies of the walk (the first version gave every thread a fixed 40-slot column: 40 KB per block, 6 waves per CU,
// 192 us on the merged 1 M-cell mesh); the finished rows leave as ONE contiguous, coalesced stream per block.
// DEG: compile-time bound of the fast path.  The generic walk is a chain of ~25 DEPENDENT global loads per row (row pointer ->
// neighbour id -> its row pointer -> its entries, one at a time: 135 us on the 1 M-cell mesh, pure latency).  A row all of
// whose faces and whose neighbours' faces number <= DEG, none of the neighbours touching a boundary, is walked branch-free in
// three waves of independent loads instead (padded slots re-read a valid entry with a zero weight; +0.0 leaves an accumulator
// unchanged, so both paths give bitwise the same sums in the same order).
template <int DEG>
__global__ void __launch_bounds__(SQN_THREADS) k_sq_numeric(
    int n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_nb, const double* __restrict__ w,
    const int32_t* __restrict__ ptr2, const int32_t* __restrict__ col2, const int32_t* __restrict__ pair_ptr,
    const uint8_t* __restrict__ slots, const uint8_t* __restrict__ fast_ok, FaceRec* __restrict__ rec2, double* __restrict__ w2,
    const int32_t* __restrict__ ell_pos = nullptr /* sliced layout of the tiled pass: CSR entry -> index in w2 (host::build_ell) */) {
  extern __shared__ double s_acc[];
  const int c0 = blockIdx.x * SQN_THREADS, c1 = min(c0 + SQN_THREADS, n);
  const int base = ptr2[c0], nloc = ptr2[c1] - base;
  for (int i = threadIdx.x; i < nloc; i += SQN_THREADS) s_acc[i] = 0.0;
  __syncthreads();
  const int c = c0 + threadIdx.x;
  if (c < c1) {
    const int o2 = ptr2[c], len2 = ptr2[c + 1] - o2;
    double* acc = s_acc + (o2 - base);
    const int pi0 = pair_ptr[c];
    const int j0 = ptr[c], deg = ptr[c + 1] - j0;
    // fast_ok (host, from the topology): 0 < deg <= DEG, a J^2 row of its own, at least one real neighbour, and every real
    // neighbour's row has <= DEG entries and no ghost face (so that its r-th entry is its r-th product)
    const bool fast = fast_ok[c] != 0;
    if (fast) {
      int mm[DEG], p0[DEG], ln[DEG], off[DEG];
      double wj[DEG];
#pragma unroll
      for (int v = 0; v < DEG / 2; ++v) {                                    // the own row, two entries per load
        const int2_u m2 = *reinterpret_cast<const int2_u*>(ent_nb + j0 + 2 * v);
        const double2_u w2v = *reinterpret_cast<const double2_u*>(w + j0 + 2 * v);
        const bool live0 = (2 * v < deg) & (m2.x >= 0), live1 = (2 * v + 1 < deg) & (m2.y >= 0);
        wj[2 * v] = live0 ? w2v.x : 0.0;     mm[2 * v] = live0 ? m2.x : -1;
        wj[2 * v + 1] = live1 ? w2v.y : 0.0; mm[2 * v + 1] = live1 ? m2.y : -1;
      }
      int run = 0;
#pragma unroll
      for (int q = 0; q < DEG; ++q) {
        const int m = mm[q] >= 0 ? mm[q] : c;
        const int2_u pp = *reinterpret_cast<const int2_u*>(ptr + m);         // (ptr[m], ptr[m + 1]) in one load
        p0[q] = pp.x;
        ln[q] = pp.y - pp.x;
        if (mm[q] >= 0) { off[q] = run; run += ln[q]; }
        else { off[q] = 0; ln[q] = 1; p0[q] = j0; }                       // dead slot: own first entry, zero weight
      }
      {
#pragma unroll
        for (int q = 0; q < DEG; ++q) {
          // the neighbour's whole row in DEG / 2 16-byte loads and its slots in one or two 4-byte loads (8- and 1-byte aligned:
          // gfx950 global loads need no natural alignment); what lies behind the row's end is read and not used -- both
          // arrays are allocated SQN_PAD elements longer.  The kernel is bound by the address rate of its gathers (about
          // 100 scalar loads per row before), not by bytes.
          double ww[DEG]; int sl[DEG];
#pragma unroll
          for (int v = 0; v < DEG / 2; ++v) {
            const double2_u t = *reinterpret_cast<const double2_u*>(w + p0[q] + 2 * v);
            ww[2 * v] = t.x; ww[2 * v + 1] = t.y;
          }
#pragma unroll
          for (int v = 0; v < (DEG + 3) / 4; ++v) {
            const uint32_t t = *reinterpret_cast<const uint32_u*>(slots + pi0 + off[q] + 4 * v);
#pragma unroll
            for (int b = 0; b < 4; ++b) if (4 * v + b < DEG) sl[4 * v + b] = (int)((t >> (8 * b)) & 255u);
          }
#pragma unroll
          for (int r = 0; r < DEG; ++r) {
            const double v = (r < ln[q]) ? wj[q] * ww[r] : 0.0;
            acc[min(sl[r], len2 - 1)] += v;
          }
        }
      }
    }
    if (!fast) {
      int pi = pi0;
      const int j1 = j0 + deg;
      for (int j = j0; j < j1; ++j) {
        const int m = ent_nb[j];
        if (m < 0) continue;
        const double wj = w[j];
        const int i1 = ptr[m + 1];
        for (int i = ptr[m]; i < i1; ++i) {
          if (ent_nb[i] < 0) continue;
          acc[(int)slots[pi]] += wj * w[i];             // (products of one row in a fixed order: deterministic)
          ++pi;
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nloc; i += SQN_THREADS) {
    const double v = s_acc[i];
    if (w2) w2[ell_pos ? ell_pos[base + i] : base + i] = v;
    if (rec2) { FaceRec out; out.nb = col2[base + i]; out.a_c = 0.0f; out.d = v; rec2[base + i] = out; }
  }
}

// Numeric J^2 on the static pattern (ptr2, col2, row2) built once on the host: one thread per OUTPUT entry (c, k);
// products J[c,m] J[m,k] are summed in a fixed order (faces of c ascending, then faces of m ascending): deterministic.
// Threads of one row are adjacent and re-read the same few adjacency rows: L1/L2 hits.
__global__ void __launch_bounds__(BLOCK) k_build_sq(int nnz2, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_nb,
                                                  const double* __restrict__ w, const int32_t* __restrict__ row2,
                                                  const int32_t* __restrict__ col2, FaceRec* __restrict__ rec2,
                                                  double* __restrict__ w2, const int32_t* __restrict__ ell_pos = nullptr) {
  const int s = blockIdx.x * BLOCK + threadIdx.x;
  if (s >= nnz2) return;
  const int c = row2[s], k = col2[s];
  double acc = 0.0;
  const int j1 = ptr[c + 1];
  for (int j = ptr[c]; j < j1; ++j) {
    const int m = ent_nb[j];
    if (m < 0) continue;
    const double wj = w[j];
    const int i1 = ptr[m + 1];
    for (int i = ptr[m]; i < i1; ++i)
      if (ent_nb[i] == k) acc += wj * w[i];
  }
  FaceRec out; out.nb = k; out.a_c = 0.0f; out.d = acc;
  rec2[s] = out;
  if (w2) w2[ell_pos ? ell_pos[s] : s] = acc;       // compact copy for the tiled pass (which has its own index array)
}

// (round 6 A/B, CWR_TCL_POWER=1) numeric J on the merged pattern of host::symbolic_j: entry (c, k) = the sum of the Jacobi weights of
// the faces between c and k, in face order.  One thread per row (rows have 4-8 entries).
__global__ void __launch_bounds__(BLOCK) k_j_numeric(int n, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent_nb, const double* __restrict__ w,
                                                    const int32_t* __restrict__ ptr2, const int32_t* __restrict__ col2, FaceRec* __restrict__ rec2,
                                                    double* __restrict__ w2, const int32_t* __restrict__ ell_pos = nullptr) {
  const int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= n) return;
  const int j0 = ptr[c], j1 = ptr[c + 1];
  for (int q = ptr2[c]; q < ptr2[c + 1]; ++q) {
    const int k = col2[q];
    double acc = 0.0;
    for (int j = j0; j < j1; ++j) if (ent_nb[j] == k) acc += w[j];
    if (w2) w2[ell_pos ? ell_pos[q] : q] = acc;
    if (rec2) { FaceRec out; out.nb = k; out.a_c = 0.0f; out.d = acc; rec2[q] = out; }
  }
}

// J^2 pass with the x tile in LDS, software-pipelined.  The plain J^2 pass (k_apply<.,5>) gathers 9-10 neighbour rows
// per row from L2 and is bound by that L2->CU gather rate and its latency, not by HBM.  Here every tile of TR rows
// comes with the list of the DISTINCT x rows its J^2 rows touch (built once on the host: own rows first, then the
// others ascending; ~2.7 rows per row for 32-row tiles under a Hilbert numbering instead of 10).  A block keeps
// three tiles in flight: it computes tile i from LDS while the x rows / weights / c2 rows of tile i+1 and the row
// list of tile i+2 are arriving in registers -- the compute phase itself issues no global load (vmcnt retires in
// order, so a load issued during compute would wait behind the prefetches).
// Compile-time prefetch depths of the tiled pass (registers holding the NEXT tile while the current one is computed):
// XR x rows per lane group (distinct x rows per tile <= XR * R), WRN J^2 entries per thread (entries per tile <= WRN * BLOCK),
// UT rows of the tile per lane group (tile rows <= UT * R).  The engine picks the cheapest configuration that fits.
struct TclCfg { int wrn, ut, xr; };
constexpr int TCL_NCFG = 10;
constexpr TclCfg TCL_CFG[TCL_NCFG] = {{4, 2, 6},      // wide rows (K = 16: 64-row tiles)
                               {10, 1, 3},     // narrow rows (K <= 8: one row per lane group, 64-256-row tiles)
                               {10, 4, 8},     // large tiles
                               {4, 1, 6},     // wide rows, four constituents per lane in the compute phase (K = 16: 4 lanes per
                                               // row, 64 rows per pass); x rows fetched 8 lanes per row, 6 per lane group
                               {4, 1, 7},      // the same with 7 x rows per lane group: halo tiles of partitioned engines may touch
                                               // a few more than 192 distinct rows (200 on one of 8 ranks of the 1 M-cell mesh)
                               {6, 1, 6},      // room for 1536 entries per tile (K = 8: 128-row tiles; 1280 overflow on meshes with 6-sided cells)
                               {6, 1, 8},      // both, 8 x rows per lane group: last resort before the un-tiled pass
                               {5, 2, 9},      // K = 20 ... 28 with TWO rows per lane group (84-102-row tiles instead of 42-51)
                               {5, 2, 12},     // K = 32: 64-row tiles (two passes of 32 rows); 16 lanes fetch a row, 12 rows each
                               {12, 1, 3}};    // narrow rows on meshes with 5-8-face cells: 256-row tiles hold up to 3 072 J^2 entries
constexpr int TCL_NARROW[4] = {0, 1, 9, 2};    // narrow-row configurations (VW <= 2), cheapest first
// Work items (long rows).  Unstructured meshes have a few cells with 5-8 faces whose J^2 rows hold 15-40 entries among
// rows of 9-10: with one lane group per ROW every wave waits for its longest row (at one constituent per lane a wave
// covers 64 rows, and 96 % of the waves of the 5 %-merged 1 M-cell mesh contain an 18-entry row: 43 us per pass against
// 27 us on the quad grid).  So a lane group works on an ITEM of at most TCL_SEG entries: item r < NR is the first
// TCL_SEG entries of the tile's row r; the remaining chunks of longer rows are extra ("virtual") items NR .. NR+NV-1 listed
// by the host (vtab: row | chunk << 8, the chunks of a row consecutive).  A virtual item leaves its partial sum in LDS
// (s_part), and after one more barrier the row's own lane group adds the partials in chunk order:
//     y = c2 + ((chunk_0 + chunk_1) + chunk_2 ...)        -- k_apply<.,5> sums in the same association (bitwise equal rows).
// Tiles therefore hold a variable number of rows (trow[t] .. trow[t+1]): rows + virtual items <= the lane groups of a pass.
static_assert(TCL_SEG % CWR_FACE_BATCH == 0, "k_apply<.,5> closes a chunk only at a face-batch boundary");
#ifndef CWR_WORK_ITEMS
#define CWR_WORK_ITEMS 0          // 1: compile the work-item (long-row splitting) path of the tiled pass; enable with CWR_TCL_SPLIT=1
#endif
#ifndef CWR_TCL_WAVES
#define CWR_TCL_WAVES 1           // __launch_bounds__ second argument of the tiled pass (A/B builds: 5 forces <= 96 VGPRs)
#endif
// ELL (round 6): the tile's entries arrive in the wave-sliced layout of host::build_ell instead of CSR order -- `ptr2` is then the per-TILE
// entry offset array (eptr), `meta` the per-tile slice offsets ((TCL_U * BLOCK / 64 + 1) per tile), and the gather loop of a wave is a scalar
// loop over the slice's padded length with constant address increments: no per-lane bounds, no exec masking (see build_ell).  G must be a
// power of two (a row's lanes never straddle two waves).  Same sums in the same order as the CSR form (padding adds + 0 x): bitwise equal.
template <int VW, int WRN, int TCL_U, int TCL_XR, bool ELL = false>
__global__ void __launch_bounds__(BLOCK, CWR_TCL_WAVES) k_sq_tiled(
    int K, int G, int TR, int ntiles, const int32_t* __restrict__ tile_list, int sched_depth, int inplace, const int32_t* __restrict__ trow,
    const int32_t* __restrict__ ptr2, const uint16_t* __restrict__ loc2,
    const double* __restrict__ w2, const int32_t* __restrict__ tcl_ptr, const int32_t* __restrict__ tcl_cols,
    const int32_t* __restrict__ vptr, const int32_t* __restrict__ meta,
    int max_cols, int stage_cap, int reps, int seg, int nvmax, const double* __restrict__ xin, const double* __restrict__ c2, double* __restrict__ yout,
    const int32_t* __restrict__ scols, int own_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  double* s_xt = reinterpret_cast<double*>(s_dyn);                                  // [max_cols][K]
  double* s_own = s_xt + (size_t)max_cols * K;                                       // [own_cap][K] the previous tile's results (reuse mode)
  double* s_part = s_own + (size_t)own_cap * K;                                      // [nvmax][K] partial sums of virtual items
  double* s_w = s_part + (size_t)nvmax * K;                                          // [stage_cap]
  uint16_t* s_loc = reinterpret_cast<uint16_t*>(s_w + stage_cap);                    // [stage_cap] (stage_cap is even)
  int32_t* s_ptr = reinterpret_cast<int32_t*>(s_loc + stage_cap);                    // [TR + 1]
  // (the codes of the tile's virtual items follow the row pointers in s_ptr: s_ptr[NR + 1 + v], sorted by row)
  const int R = BLOCK / G;
  const int tid = threadIdx.x;
  const int r = tid / G, g = tid - r * G;
  const int col = g * VW;
  // VW == 4 (wide rows): a lane owns two 16-byte pieces of its row, {2g, 2g+1} and {K/2 + 2g, K/2 + 2g + 1}, and the odd
  // row slots of a wave hold them in swapped register order.  The 8 lanes that an LDS b128 read serves together then
  // cover all 32 banks (low half of one row, high half of the next) instead of hitting the same 16 banks twice.
  const int offA = (VW == 4) ? ((r & 1) ? K / 2 + 2 * g : 2 * g) : col;
  const int offB = (VW == 4) ? ((r & 1) ? 2 * g : K / 2 + 2 * g) : col;
  auto ld_row = [&](const double* rowp, double (&v)[VW]) {
    if constexpr (VW == 4) {
      const double2 a = *reinterpret_cast<const double2*>(rowp + offA), b = *reinterpret_cast<const double2*>(rowp + offB);
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else ldv<VW>(rowp + col, v);
  };
  auto ld_row_nt = [&](const double* rowp, double (&v)[VW]) {
    if constexpr (VW == 4) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      const d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2*>(rowp + offA));
      const d2 b = __builtin_nontemporal_load(reinterpret_cast<const d2*>(rowp + offB));
      v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else ldv_nt<VW>(rowp + col, v);
  };
  auto st_row = [&](double* rowp, const double (&v)[VW]) {
    if constexpr (VW == 4) {
      *reinterpret_cast<double2*>(rowp + offA) = make_double2(v[0], v[1]);
      *reinterpret_cast<double2*>(rowp + offB) = make_double2(v[2], v[3]);
    } else stv<VW>(rowp + col, v);
  };
  // sum over the entries [j0, j1) of (weight x the x row at the entry's in-tile position); s_loc holds the position
  // pre-multiplied by K (a double index into s_xt: no multiply in the loop).  Entries ascending from 0.0, as k_apply<.,5>.
  // (Measured and dropped, round 2: a two-entry software pipeline of this loop -- index two entries ahead, row one ahead,
  // sched_barriers against hipcc's re-sinking -- makes the EXACT pass faster, 97.7 -> 91.4 us at K = 16, because there the
  // two dependent LDS round trips per entry are exposed; but it re-reads one row per row and pads odd rows, and with two
  // or three tile-local applications the phase is LDS-bandwidth-bound (LDS busy 50 % of the pass, a quarter of it bank
  // conflicts of the 64-byte row pieces): 109.7 -> 117.1 us and 135 -> 150 us.  profiles/r02_k_pipelined_loop.txt.
  // Likewise a quad-broadcast fetch -- the four lanes of a row read the (weight, position) pairs of four consecutive entries
  // and pass them round with DPP, a quarter of the LDS instructions for them: exact pass 95.7 -> 91.6 us, three applications
  // 135 -> 131 us, but the default two applications 103.5 -> 106.9 us (three DPP moves and an exec branch per entry).
  // And two entries per trip without padding (odd entry handled after the loop): 105 -> 139 us at K = 16, 65 -> 89 at K = 8,
  // 38 -> 57 at K = 1 (same box, alternating libraries, scratch/r02_ab.sh): the one-entry loop stays.
  // And the lightest form -- only the NEXT entry's (position, weight) asked for one trip ahead, through inline asm because
  // hipcc rotates a C version back: 104.2 -> 109.3 us at K = 16, 90.0 -> 92.6 at K = 12, 65.3 -> 67.0 at K = 8, 30.6 -> 31.1
  // at K = 1.  The two dependent LDS round trips per entry are not what the pass waits for.)
  auto row_sum = [&](int j0, int j1, double (&sum)[VW]) {
#pragma unroll
    for (int w = 0; w < VW; ++w) sum[w] = 0.0;
    for (int j = j0; j < j1; ++j) {
      double xn[VW];
      ld_row(s_xt + (int)s_loc[j], xn);
      const double wj = s_w[j];
#pragma unroll
      for (int w = 0; w < VW; ++w) sum[w] += wj * xn[w];
    }
  };
  // SPLIT: the work-item logic, compiled into the one-constituent-per-lane variants of a -DCWR_WORK_ITEMS=1 build only.  It is
  // NOT part of the default build: once 256-row tiles fit without it (configuration {12, 1, 3}) it gains 1 % at K = 1 in a
  // same-box comparison, less than the tile-balanced numbering gives for free (3 %), and it costs 12 VGPRs and a barrier
  // wherever rows share a lane's loop (K = 16: 110 -> 119 us).  profiles/r02_r_k1_ab.txt
  constexpr bool SPLIT = (VW == 1) && (CWR_WORK_ITEMS != 0) && !ELL;
  constexpr int NSL = TCL_U * (BLOCK / 64);          // ELL: slices of a tile (one per wave and row set)
  const bool rowlane = r < R;
  // x rows are FETCHED (global -> registers -> LDS) one whole 128-byte row per 8 lanes whatever VW is: with VW == 4 the
  // compute mapping above would fetch half rows (measured +6 us per pass)
  constexpr int XW = (VW == 4) ? 2 : VW;
  const int GL = (VW == 4) ? K / 2 : G, RL = BLOCK / GL;
  const int rl = tid / GL, gl = tid - rl * GL;
  const bool loadlane = rl < RL;
  const int xcd = blockIdx.x % N_XCD, lidx = blockIdx.x / N_XCD, bpx = gridDim.x / N_XCD;
  const int tpx = (ntiles + N_XCD - 1) / N_XCD;
  auto tile_of = [&](int it) -> int {              // it-th tile of this block, or -1
    if (sched_depth > 0) {
      // chained pass: tile_list is a SCHEDULE [sched_depth][gridDim.x] -- every block walks its own list of tiles (chains of
      // tiles linked along the flow, two of them interleaved: see cwr_engine.hip build_chain_schedule), -1 = end of the list
      if (it >= sched_depth) return -1;
      return __builtin_amdgcn_readfirstlane(tile_list[(size_t)it * gridDim.x + blockIdx.x]);
    }
    const int i = lidx + it * bpx;
    if (i >= tpx) return -1;
    const int t = xcd * tpx + i;
    if (t >= ntiles) return -1;
    return __builtin_amdgcn_readfirstlane(tile_list ? tile_list[t] : t);   // (a launch may cover a subset of the tiles: interior / cut tiles)
  };
  // prefetch registers
  int cn[TCL_XR];                                  // row list of the tile after next (global x row ids)
  // REUSE mode (scols != nullptr; chained passes): the column list of a tile comes from a per-schedule copy in which a column that
  // the PREVIOUS tile of this block's list already holds in LDS is coded -2 - (its position there) instead of its row id.
  // Such a column is not fetched: it is carried over from the old LDS image -- for the previous tile's own rows from s_own, i.e.
  // its fresh results.  With lane-major numbering consecutive tiles of a list overlap in 64 of their 152 columns (the upstream
  // halo = the predecessor's last two columns, and the own rows that were the predecessor's downstream halo).
  // (positions travel as 16-bit halves, two to a register: the codes of the current tile cost (XR + 1) / 2 VGPRs instead of XR;
  // 0xFFFF = fetched, not carried over)
  constexpr int NCC = (TCL_XR + 1) / 2;
  uint32_t cc[NCC];                                // the carry-over positions of the CURRENT tile's columns
  auto save_codes = [&]() {
#pragma unroll
    for (int w = 0; w < NCC; ++w) cc[w] = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < TCL_XR; ++u) {
      const uint32_t b = (cn[u] <= -2) ? (uint32_t)(-2 - cn[u]) : 0xFFFFu;
      cc[u >> 1] = (cc[u >> 1] & ~(0xFFFFu << (16 * (u & 1)))) | (b << (16 * (u & 1)));
    }
  };
  double xr[TCL_XR][XW];                           // x rows of the next tile (fetch mapping)
  double wr[WRN]; int lr[WRN];               // weights / local indices of the next tile
  double q0[TCL_U][VW];                            // c2 rows of the next tile
  int pr = 0;                                      // row pointer slice of the next tile (then the codes of its virtual items)
  // (row range, virtual-item range) of: the tile after next (a: loaded with its row list), the next tile (n), the current (c)
  int a_c0 = 0, a_c1 = 0, a_v0 = 0, a_v1 = 0, n_c0 = 0, n_c1 = 0, n_v0 = 0, n_v1 = 0;
  auto load_cols = [&](int t) {
#pragma unroll
    for (int u = 0; u < TCL_XR; ++u) cn[u] = -1;
    a_c0 = a_c1 = a_v0 = a_v1 = 0;
    if (t < 0) return;
    a_c0 = __builtin_amdgcn_readfirstlane(trow[t]); a_c1 = __builtin_amdgcn_readfirstlane(trow[t + 1]);     // (wave-uniform: SGPRs)
    a_v0 = __builtin_amdgcn_readfirstlane(vptr[t]); a_v1 = __builtin_amdgcn_readfirstlane(vptr[t + 1]);
    if (!loadlane) return;
    const int cb = tcl_ptr[t], ce = tcl_ptr[t + 1];
    const int32_t* cols = scols ? scols : tcl_cols;
#pragma unroll
    for (int u = 0; u < TCL_XR; ++u) { const int q = rl + u * RL; if (q < ce - cb) cn[u] = cols[cb + q]; }
  };
  auto load_rows = [&](int t) {                    // uses cn and (n_c0, n_c1, n_v0, n_v1), loaded one tile earlier
    if (t < 0) return;
#pragma unroll
    for (int u = 0; u < TCL_XR; ++u) if (cn[u] >= 0) ldv<XW>(xin + (size_t)cn[u] * K + gl * XW, xr[u]);
    const int c0 = n_c0, c1 = n_c1;
    const int jb = ELL ? ptr2[t] : ptr2[c0], je = ELL ? ptr2[t + 1] : ptr2[c1];
#pragma unroll
    for (int u = 0; u < WRN; ++u) { const int j = jb + tid + u * BLOCK; if (j < je) {
      // wide rows (the four-wide mapping): the entry stream bypasses cache retention so that the gathered x rows keep the L2
      // (97 vs 103 us per pass at K = 16); narrow rows lose from it (32 vs 26 us at K = 1: the entries ARE their traffic)
      if constexpr (VW == 4) { wr[u] = __builtin_nontemporal_load(w2 + j); lr[u] = __builtin_nontemporal_load(loc2 + j); }
      else { wr[u] = w2[j]; lr[u] = loc2[j]; }
    } }
    // raw: rebased when it is stored to LDS (a subtraction here would wait for this load, and with it for every prefetch
    // issued before it).  The threads behind the tile's rows fetch the codes of its virtual items through the same load.
    // (`meta`: per tile its rows' ptr2 entries followed by the codes of its virtual items, tiles back to back -- one
    // uniform base + tid, exactly the shape of the row-pointer prefetch it replaces)
    if constexpr (ELL) { if (tid <= NSL) pr = meta[t * (NSL + 1) + tid]; }        // (the tile's slice offsets)
    else if (tid < (c1 - c0) + (n_v1 - n_v0)) pr = meta[c0 + n_v0 + tid];
    if (rowlane) {
#pragma unroll
      for (int u = 0; u < TCL_U; ++u) { const int c = c0 + r + u * R; if (c < c1) ld_row_nt(c2 + (size_t)c * K, q0[u]); }
    }
  };
  int t_cur = tile_of(0);
  load_cols(t_cur);
  n_c0 = a_c0; n_c1 = a_c1; n_v0 = a_v0; n_v1 = a_v1;
  load_rows(t_cur);                                // (the first tile's chain is not hidden)
  save_codes();
  int c_c0 = n_c0, c_c1 = n_c1, c_nv = n_v1 - n_v0;
  int t_next = tile_of(1);
  load_cols(t_next);
  n_c0 = a_c0; n_c1 = a_c1; n_v0 = a_v0; n_v1 = a_v1;
  // Results are stored one tile LATE, right before the next tile's prefetch is issued: on gfx9 stores count in vmcnt like
  // loads, so a store issued at the end of a tile would be the youngest entry the next tile's `wait for the prefetch` has
  // to drain (a full write latency per tile); deferred, it has the whole compute phase to retire.
  double y[TCL_U][VW];
  int pc0 = 0, pc1 = 0;                            // rows of the tile whose results y still holds
  for (int it = 0; t_cur >= 0; ++it) {
    const int c0 = c_c0, c1 = c_c1, NR = c1 - c0, nv = c_nv;
    const int ncol = tcl_ptr[t_cur + 1] - tcl_ptr[t_cur];
    const int jb0 = ELL ? ptr2[t_cur] : ptr2[c0];
    const int nent = (ELL ? ptr2[t_cur + 1] : ptr2[c1]) - jb0;
    __syncthreads();                               // the previous tile's readers are done with LDS
    if (scols && loadlane) {
      // carry over the previous tile's own rows -- the results it has just computed -- from the staging area (written before the
      // barrier above; the columns it held as halo were picked up from the old image at the end of its compute phase, below:
      // nothing reads the old image any more, so no second barrier is needed before it is overwritten)
      const int pNR = pc1 - pc0;
#pragma unroll
      for (int u = 0; u < TCL_XR; ++u) {
        const int pos = (int)((cc[u >> 1] >> (16 * (u & 1))) & 0xFFFFu);
        if (pos < pNR) ldv<XW>(s_own + (size_t)pos * K + gl * XW, xr[u]);
      }
    }
    if (loadlane) {
#pragma unroll
      for (int u = 0; u < TCL_XR; ++u) { const int q = rl + u * RL; if (q < ncol) stv<XW>(s_xt + (size_t)q * K + gl * XW, xr[u]); }
    }
#pragma unroll
    for (int u = 0; u < WRN; ++u) { const int j = tid + u * BLOCK; if (j < nent) { s_w[j] = wr[u]; s_loc[j] = (uint16_t)lr[u]; } }
    // one store for both kinds: row pointers rebased to the tile, virtual-item codes as they are, behind the terminator
    // (a second, separate store for the codes cost this kernel 26 VGPRs -- a block per CU -- in hipcc's allocation)
    if constexpr (ELL) { if (tid <= NSL) s_ptr[tid] = pr; }
    else {
    if (tid < NR + nv) s_ptr[tid + (tid >= NR ? 1 : 0)] = pr - (tid < NR ? jb0 : 0);
    if (tid == 0) s_ptr[NR] = nent;
    }
    double qc[TCL_U][VW];
#pragma unroll
    for (int u = 0; u < TCL_U; ++u)
#pragma unroll
      for (int w = 0; w < VW; ++w) qc[u][w] = q0[u][w];
    // in-place passes (xin == yout, chained schedule): the previous tile's results leave BEFORE the barrier, so that every
    // wave's stores are issued before any wave's prefetch of the next tile's x rows below -- the tile after next in this
    // block's list is the chain successor of the previous one and must read what it has just written (same CU, same L1:
    // workgroup-scope ordering needs no cache action on gfx950, and hipcc's barrier carries no vmcnt wait)
    // (reuse mode needs neither: the successor takes these rows from LDS)
    const bool early = inplace && !scols;
    if (early && rowlane) {
#pragma unroll
      for (int u = 0; u < TCL_U; ++u) { const int c = pc0 + r + u * R; if (c < pc1) st_row(yout + (size_t)c * K, y[u]); }
    }
    __syncthreads();
    if (!early && rowlane) {                       // the previous tile's results
#pragma unroll
      for (int u = 0; u < TCL_U; ++u) { const int c = pc0 + r + u * R; if (c < pc1) st_row(yout + (size_t)c * K, y[u]); }
    }
    // (Measured, round 3: an acquire fence in front of this prefetch changes nothing at workgroup scope -- 43-47 sweeps either
    // way: the chain successor already reads its predecessor's rows fresh -- and at agent scope, which invalidates the L2,
    // the same sweeps take 3.5 x the time: gpurun_out/r03k_acquire.txt.)
    // start the next tile's loads (x rows by the list already in registers) and the list of the tile after it
    const int t_after = tile_of(it + 2);
    load_rows(t_next);
    if (scols) save_codes();                                    // (the next tile's carry-over positions, for its LDS image one iteration on)
    const int x_c0 = n_c0, x_c1 = n_c1, x_nv = n_v1 - n_v0;      // (the next tile's ranges, before load_cols overwrites a_*)
    load_cols(t_after);
    n_c0 = a_c0; n_c1 = a_c1; n_v0 = a_v0; n_v1 = a_v1;
    // reps > 1 (block-asynchronous Jacobi): the tile applies J^2 again to its own freshly computed rows, which sit first
    // in the LDS x tile, while rows of other tiles keep the values of the pass's input.  A chaotic relaxation in the
    // sense of Chazan-Miranker: it converges whenever rho(|J|) < 1 (always here: A is a strictly diagonally dominant
    // M-matrix) and the inner applications cost LDS reads only.
    for (int rep = 0; rep < reps; ++rep) {
      if (rep > 0) {
        __syncthreads();                             // every reader of the previous round is done
        if (rowlane) {
#pragma unroll
          for (int u = 0; u < TCL_U; ++u) { const int i = r + u * R; if (i < NR) st_row(s_xt + (size_t)i * K, y[u]); }
        }
        __syncthreads();
      }
      if constexpr (ELL) {
        // wave-sliced entries: every lane of a wave makes the slice's L trips (padding: weight 0 on the row's own cell), the k-th gather
        // of the wave reads rpw consecutive entries -- a scalar loop, addresses advancing by a constant
        const int rsh = 6 - __builtin_ctz((unsigned)G);                  // log2(rows per wave); G a power of two
        const int r16 = (tid & 63) >> __builtin_ctz((unsigned)G);        // the lane's row within its wave's slice
#pragma unroll
        for (int u = 0; u < TCL_U; ++u) {
          const int sidx = u * (BLOCK / 64) + (tid >> 6);
          const int sb = __builtin_amdgcn_readfirstlane(s_ptr[sidx]);
          const int L = __builtin_amdgcn_readfirstlane((s_ptr[sidx + 1] - sb) >> rsh);
          double sum[VW];
#pragma unroll
          for (int w = 0; w < VW; ++w) sum[w] = 0.0;
          const uint16_t* pl = s_loc + sb + r16;
          const double* pw = s_w + sb + r16;
          const int stride = 1 << rsh;
          for (int k = 0; k < L; ++k) {
            double xn[VW];
            ld_row(s_xt + (int)pl[0], xn);
            const double wj = pw[0];
#pragma unroll
            for (int w = 0; w < VW; ++w) sum[w] += wj * xn[w];
            pl += stride; pw += stride;
          }
#pragma unroll
          for (int w = 0; w < VW; ++w) y[u][w] = qc[u][w] + sum[w];     // (lanes behind the tile's last row: zero sums, never stored)
        }
      } else if constexpr (!SPLIT) {
        // one lane group per row (every item is a whole row: the host lists no virtual items for these variants)
        if (rowlane) {
#pragma unroll
          for (int u = 0; u < TCL_U; ++u) {
            const int i = r + u * R;
            if (i < NR) {
              double sum[VW];                          // same association as k_apply<.,5>: (sum of w x) first, + c2 last,
              row_sum(s_ptr[i], s_ptr[i + 1], sum);    // so both J^2 kernels give bitwise equal rows
#pragma unroll
              for (int w = 0; w < VW; ++w) y[u][w] = qc[u][w] + sum[w];
            }
          }
        }
      } else {
      int jfull[TCL_U];                              // entries of the item's whole row (real items)
      if (rowlane) {
#pragma unroll
        for (int u = 0; u < TCL_U; ++u) {
          const int i = r + u * R;                   // item
          jfull[u] = 0;
          if (i < NR + nv) {
            int row = i, j0, jend;
            if (i < NR) { j0 = s_ptr[i]; jend = s_ptr[i + 1]; jfull[u] = jend - j0; }
            else { const int code = s_ptr[i + 1]; row = code & 255; j0 = s_ptr[row] + (code >> 8) * seg; jend = s_ptr[row + 1]; }
            const int j1 = min(jend, j0 + seg);
            double sum[VW];                          // same association as k_apply<.,5>: (sum of w x) per chunk, + c2 last,
            row_sum(j0, j1, sum);                    // so both J^2 kernels give bitwise equal rows
            if (i < NR) {
#pragma unroll
              for (int w = 0; w < VW; ++w) y[u][w] = sum[w];
            } else st_row(s_part + (size_t)(i - NR) * K, sum);
          }
        }
      }
      if (nv > 0) __syncthreads();                   // (uniform per tile) the partial sums of the long rows are in LDS
      if (rowlane) {
#pragma unroll
        for (int u = 0; u < TCL_U; ++u) {
          const int i = r + u * R;
          if (i < NR) {
            if (jfull[u] > seg) {
              const int nc = (jfull[u] - 1) / seg;
              int lo = 0, hi = nv;                     // first virtual item of row i (the codes are sorted by row)
              while (lo < hi) { const int mid = (lo + hi) >> 1; if ((s_ptr[NR + 1 + mid] & 255) < i) lo = mid + 1; else hi = mid; }
              const int fv = lo;
              for (int q = 0; q < nc; ++q) {
                double t[VW];
                ld_row(s_part + (size_t)(fv + q) * K, t);
#pragma unroll
                for (int w = 0; w < VW; ++w) y[u][w] += t[w];
              }
            }
#pragma unroll
            for (int w = 0; w < VW; ++w) y[u][w] = qc[u][w] + y[u][w];
          }
        }
      }
          }
    }
    if (scols) {
      if (rowlane) {                               // the results, where the next tile of this list looks for them
#pragma unroll
        for (int u = 0; u < TCL_U; ++u) { const int i = r + u * R; if (i < NR) st_row(s_own + (size_t)i * K, y[u]); }
      }
      if (loadlane) {
        // ... and the halo columns of THIS tile that the next one needs too, from the image while it is still there (those rows
        // are read-only during the compute phase, so other waves may still be computing): into the free prefetch registers --
        // a carried-over column was not fetched.  cc holds the next tile's codes since its prefetch was issued.
#pragma unroll
        for (int u = 0; u < TCL_XR; ++u) {
          const int pos = (int)((cc[u >> 1] >> (16 * (u & 1))) & 0xFFFFu);
          if (pos != 0xFFFF && pos >= NR) ldv<XW>(s_xt + (size_t)pos * K + gl * XW, xr[u]);
        }
      }
    }
    pc0 = c0; pc1 = c1;
    t_cur = t_next;
    t_next = t_after;
    c_c0 = x_c0; c_c1 = x_c1; c_nv = x_nv;
  }
  if (rowlane) {                                   // the last tile's results
#pragma unroll
    for (int u = 0; u < TCL_U; ++u) { const int c = pc0 + r + u * R; if (c < pc1) st_row(yout + (size_t)c * K, y[u]); }
  }
}

// Flow between tiles (chained passes, see cwr_engine.hip build_chain_schedule): one thread per DIRECTED tile link sums, over the
// faces through which cells of the link's source tile meet cells of its destination tile, the flow LEAVING the source side
// at this time level (entries: face index << 1 | side; side 1 = the source cell is face2, so its outflow is -advection_coeff).
// The entries of a link are contiguous and summed in a fixed order: the schedule built from these sums is reproducible.
__global__ void __launch_bounds__(BLOCK) k_link_flux(int n_links, const int32_t* __restrict__ link_ptr, const int32_t* __restrict__ link_ent,
                                                   const float* __restrict__ adv_t, float* __restrict__ flux) {
  const int l = blockIdx.x * BLOCK + threadIdx.x;
  if (l >= n_links) return;
  double s = 0.0;
  const int j1 = link_ptr[l + 1];
  for (int j = link_ptr[l]; j < j1; ++j) {
    const int code = link_ent[j];
    const double a = (double)adv_t[code >> 1];
    s += fmax((code & 1) ? -a : a, 0.0);
  }
  flux[l] = (float)s;
}

// ------------------------------------------------------------------------------------------------ BiCGSTAB vector kernels
// rr_prev = the ||r||^2 row of slot (it-1)%3; a column is frozen once it is below tol^2 ||bhat||^2.
__device__ __forceinline__ bool col_active(const double* rr_prev, const double* bb, double tol2, int k) {
  return rr_prev[k] > tol2 * bb[k];
}

// s = r - alpha v,  alpha = rho / (r0, v) on active columns, 0 on converged ones.
template <int VW>
__global__ void __launch_bounds__(BLOCK) k_vec_s(
    int n_owned, int K, int G, const double* __restrict__ r, const double* __restrict__ v,
    double* __restrict__ s, const double* __restrict__ rho, const double* __restrict__ acc_cur,
    const double* __restrict__ rr_prev, const double* __restrict__ bb, double tol2) {
  const int R = BLOCK / G;
  const int rr_ = threadIdx.x / G, g = threadIdx.x - rr_ * G;
  if (rr_ >= R) return;
  double alpha[VW];
#pragma unroll
  for (int w = 0; w < VW; ++w) {
    const int k = g * VW + w;
    const double den = acc_cur[ACC_R0V * K + k];
    alpha[w] = (col_active(rr_prev, bb, tol2, k) && den != 0.0) ? rho[k] / den : 0.0;
  }
  for (int c = blockIdx.x * R + rr_; c < n_owned; c += gridDim.x * R) {
    const size_t o = (size_t)c * K + g * VW;
    double a[VW], b[VW];
    ldv<VW>(r + o, a); ldv<VW>(v + o, b);
#pragma unroll
    for (int w = 0; w < VW; ++w) a[w] -= alpha[w] * b[w];
    stv<VW>(s + o, a);
  }
}

// omega = (t,s)/(t,t); x += alpha p + omega s; r = s - omega t;
// rho' = (r0,s) - omega (r0,t)   [= (r0, r') by linearity, from two TRUE inner products of the second
// product's launch: the textbook shortcut rho - alpha (r0,v) - omega (r0,t) accumulates the rounding of
// every earlier rho and stalls the iteration once r is nearly orthogonal to r0];
// beta = (rho'/rho)(alpha/omega); p = r + beta (p - omega v); acc[RR] += (r', r').
// Block 0 also publishes rho' and keeps the counters.
template <int VW>
__global__ void __launch_bounds__(BLOCK) k_vec_x(
    int n_owned, int K, int G, double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
    const double* __restrict__ s, const double* __restrict__ t, const double* __restrict__ v,
    const double* __restrict__ rho, double* __restrict__ rho_next, const double* __restrict__ acc_cur,
    double* __restrict__ partial, const double* __restrict__ rr_prev, const double* __restrict__ bb,
    double tol2, int32_t* __restrict__ counters) {
  __shared__ double s_red[BLOCK * 2];
  const int R = BLOCK / G;
  const int rr_ = threadIdx.x / G, g = threadIdx.x - rr_ * G;
  double alpha[VW], omega[VW], beta[VW], part[VW];
#pragma unroll
  for (int w = 0; w < VW; ++w) { alpha[w] = omega[w] = beta[w] = part[w] = 0.0; }
  if (rr_ < R) {
#pragma unroll
    for (int w = 0; w < VW; ++w) {
      const int k = g * VW + w;
      const bool act = col_active(rr_prev, bb, tol2, k);
      const double r0v = acc_cur[ACC_R0V * K + k], ts = acc_cur[ACC_TS * K + k];
      const double tt = acc_cur[ACC_TT * K + k], r0t = acc_cur[ACC_R0T * K + k];
      const double r0s = acc_cur[ACC_R0S * K + k];
      const double rh = rho[k];
      const double al = (act && r0v != 0.0) ? rh / r0v : 0.0;
      const double om = (act && tt > 0.0) ? ts / tt : 0.0;
      const double rh2 = r0s - om * r0t;
      alpha[w] = al; omega[w] = om;
      beta[w] = (act && om != 0.0 && rh != 0.0) ? (rh2 / rh) * (al / om) : 0.0;
    }
    for (int c = blockIdx.x * R + rr_; c < n_owned; c += gridDim.x * R) {
      const size_t o = (size_t)c * K + g * VW;
      double xv[VW], pv[VW], sv[VW], tv[VW], vv[VW], rv[VW];
      ldv<VW>(x + o, xv); ldv<VW>(p + o, pv); ldv<VW>(s + o, sv); ldv<VW>(t + o, tv); ldv<VW>(v + o, vv);
#pragma unroll
      for (int w = 0; w < VW; ++w) {
        xv[w] += alpha[w] * pv[w] + omega[w] * sv[w];
        rv[w] = sv[w] - omega[w] * tv[w];
        pv[w] = rv[w] + beta[w] * (pv[w] - omega[w] * vv[w]);
        part[w] += rv[w] * rv[w];
      }
      stv<VW>(x + o, xv); stv<VW>(r + o, rv); stv<VW>(p + o, pv);
    }
  }
  block_reduce_cols<VW, VW>(part, G, K, partial + (size_t)blockIdx.x * K, s_red);
  if (blockIdx.x == 0) {
    const int tid = threadIdx.x;
    if (tid < K) {
      const int k = tid;
      const bool act = col_active(rr_prev, bb, tol2, k);
      const double r0v = acc_cur[ACC_R0V * K + k], ts = acc_cur[ACC_TS * K + k];
      const double tt = acc_cur[ACC_TT * K + k], r0t = acc_cur[ACC_R0T * K + k];
      const double r0s = acc_cur[ACC_R0S * K + k];
      const double rh = rho[k];
      const double al = (act && r0v != 0.0) ? rh / r0v : 0.0;
      const double om = (act && tt > 0.0) ? ts / tt : 0.0;
      const double rh2 = r0s - om * r0t;
      rho_next[k] = act ? rh2 : rh;
      if (act && (r0v == 0.0 || om == 0.0 || rh2 == 0.0)) counters[1] = 1;        // (near-)breakdown: host restarts
      if (act && !(fabs(rh2) < 1.0e300 && fabs(al) < 1.0e300)) counters[3] = 1;  // NaN / Inf
    }
    if (tid == 0) {                       // single writer: iterations in which some column still worked
      bool any = false;
      for (int k = 0; k < K; ++k) any = any || col_active(rr_prev, bb, tol2, k);
      if (any) counters[0] += 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ a-5
// mesh[name][t+1, ghost] = input_array[t+1, ghost] where non-zero, NaN otherwise
// (transport.py:258-264 over the NaN-initialised array of constituents.py:39-48).
__global__ void __launch_bounds__(BLOCK) k_ghost_writeback(int64_t total, const double* __restrict__ bc_n,
                                                         double* __restrict__ ghost_rows) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= total) return;
  const double v = bc_n[i];
  ghost_rows[i] = (v != 0.0) ? v : __builtin_nan("");
}

// Real-cell entries of input_array[t+1] (point sources / fixed concentrations inside the domain): the reference
// overwrites the solved level with every non-zero entry before the mass fluxes are taken (transport.py:258-264).
// Sparse: (row, K values) per entry of the level; zero = "no input" as everywhere in the reference.
__global__ void __launch_bounds__(BLOCK) k_apply_inputs(int64_t total, int K, const int32_t* __restrict__ rows,
                                                      const double* __restrict__ vals, double* __restrict__ c) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= total) return;
  const int64_t ent = i / K;
  const int k = (int)(i - ent * K);
  const double v = vals[i];
  if (v != 0.0) c[(size_t)rows[ent] * K + k] = v;
}

// ------------------------------------------------------------------------------------------------ a-6
// transport.py:414-429 with c = concentrations of level t+1 (ghost rows included, NaN = no value).
template <int VW>
__global__ void __launch_bounds__(BLOCK) k_mass_flux(
    int E, int n_owned, int K, int G, const int32_t* __restrict__ f1, const int32_t* __restrict__ f2,
    const float* __restrict__ adv_t, const double* __restrict__ dif_t, double dt,
    const double* __restrict__ c, double* __restrict__ fadv,
    double* __restrict__ fdif, const int32_t* __restrict__ face_list, int n_list) {
  // face_list (optional): only these faces (a partitioned engine takes the faces between core cells beside the exchange that
  // refreshes its halo rows, the others behind it)
  const int R = BLOCK / G;
  const int r = threadIdx.x / G, g = threadIdx.x - r * G;
  if (r >= R) return;
  const int n_it = face_list ? n_list : E;
  for (int it = blockIdx.x * R + r; it < n_it; it += gridDim.x * R) {
    const int e = face_list ? face_list[it] : it;                 // e: internal face index (faces sorted along the cell order)
    const int P = f1[e], N = f2[e];
    // the three arrays are written in the INTERNAL face order: 3 x 8 K E bytes leave as one sequential stream (written at the
    // reference's face ids they were a scatter of 128-byte rows: 233 vs 150 us at 1 M cells x 16); every reader
    // (cwr_get_mass_flux, the output snapshot) goes through the face map
    // (round 4) total_mass_flux = advection + diffusion (transport.py:427-429) is NOT stored: the two addends are, and every
    // reader forms the sum on the way out (cwr_get_mass_flux, the output snapshot: the same IEEE addition of the same two doubles,
    // bit for bit) -- a third of this kernel's 0.8 GB of output per step at 1 M cells x 16 was that redundant array
    const size_t o = (size_t)e * K + g * VW;
    double oa[VW], od[VW];
    if (P >= n_owned) {                 // face owned by another rank
#pragma unroll
      for (int w = 0; w < VW; ++w) oa[w] = od[w] = 0.0;
    } else {
      const float a = adv_t[e];
      const double d = dif_t[e];
      double cp[VW], cn[VW];
      ldv<VW>(c + (size_t)P * K + g * VW, cp);
      ldv<VW>(c + (size_t)N * K + g * VW, cn);
#pragma unroll
      for (int w = 0; w < VW; ++w) {
        oa[w] = ((a < 0.0f) ? (double)a * cn[w] : (double)a * cp[w]) * dt;
        od[w] = d * (cn[w] - cp[w]) * dt;
      }
    }
    stv<VW>(fadv + o, oa); stv<VW>(fdif + o, od);
  }
}

// ------------------------------------------------------------------------------------------------ 8f-4 output side
// Mass-balance ledger (postproc_util.py:84-139): one block per boundary-condition line.  For the line's faces owned by
// this engine, total_mass_flux of this step (transport.py:414-429, concentrations of level t+1, ghost rows included)
// is summed per constituent into ledger[line][0] (all), [1] (the part <= 0: into the domain) and [2] (the part >= 0);
// `f * 0` instead of 0 keeps the reference's NaN propagation (np.where(x <= 0, x, x * 0)).  Fixed summation order.
__global__ void __launch_bounds__(BLOCK) k_line_mass(
    int K, int n_core, const int32_t* __restrict__ line_ptr, const int32_t* __restrict__ line_faces,
    const int32_t* __restrict__ f1, const int32_t* __restrict__ f2, const float* __restrict__ adv_t,
    const double* __restrict__ dif_t, double dt, const double* __restrict__ c, double* __restrict__ ledger) {
  __shared__ double s_acc[3][BLOCK];
  const int line = blockIdx.x, tid = threadIdx.x;
  const int per = BLOCK / K;                       // faces in flight (K <= BLOCK checked by the host)
  double tot = 0.0, tin = 0.0, tout = 0.0;
  if (tid < per * K) {
    const int k = tid % K;
    for (int i = line_ptr[line] + tid / K; i < line_ptr[line + 1]; i += per) {
      const int e = line_faces[i];
      const int P = f1[e], N = f2[e];
      if (P >= n_core) continue;                   // another rank owns this face
      const float a = adv_t[e];
      const double d = dif_t[e];
      const double cp = c[(size_t)P * K + k], cn = c[(size_t)N * K + k];
      const double f = ((a < 0.0f) ? (double)a * cn : (double)a * cp) * dt + d * (cn - cp) * dt;
      tot += f;
      tin += (f <= 0.0) ? f : f * 0.0;
      tout += (f >= 0.0) ? f : f * 0.0;
    }
  }
  s_acc[0][tid] = tot; s_acc[1][tid] = tin; s_acc[2][tid] = tout;
  __syncthreads();
  if (tid < K) {
    for (int q = 0; q < 3; ++q) {
      double sum = 0.0;
      for (int j = 0; j < per; ++j) sum += s_acc[q][j * K + tid];
      ledger[((size_t)line * 3 + q) * K + tid] += sum;
    }
  }
}

// out[k] = sum_c vol[c] * x[c, k] over the engine's own real cells, out[K] = sum_c vol[c]  (postproc_util.py:36-57).
// Stage 1: per-block partials [grid][K + 1]; stage 2 (k_fold_partials): one block folds them in block order.
__global__ void __launch_bounds__(BLOCK) k_domain_mass(int n, int K, const float* __restrict__ vol,
                                                     const double* __restrict__ x, double* __restrict__ partial) {
  __shared__ double s_acc[BLOCK];
  __shared__ double s_vol[BLOCK];
  const int tid = threadIdx.x, per = BLOCK / K;
  double acc = 0.0, accv = 0.0;
  if (tid < per * K) {
    const int k = tid % K;
    for (int c = blockIdx.x * per + tid / K; c < n; c += gridDim.x * per) {
      const double v = (double)vol[c];
      acc += v * x[(size_t)c * K + k];
      if (k == 0) accv += v;
    }
  }
  s_acc[tid] = acc; s_vol[tid] = accv;
  __syncthreads();
  if (tid < K) {
    double sum = 0.0;
    for (int j = 0; j < per; ++j) sum += s_acc[j * K + tid];
    partial[(size_t)blockIdx.x * (K + 1) + tid] = sum;
  }
  if (tid == 0) {
    double sum = 0.0;
    for (int j = 0; j < per; ++j) sum += s_vol[j * K];
    partial[(size_t)blockIdx.x * (K + 1) + K] = sum;
  }
}

__global__ void __launch_bounds__(BLOCK) k_fold_partials(int nblocks, int width, const double* __restrict__ partial,
                                                       double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= width) return;
  double sum = 0.0;
  for (int b = 0; b < nblocks; ++b) sum += partial[(size_t)b * width + k];
  out[k] = sum;
}

// Output snapshot, constituent-major: out[k * n_out + i] = x[row(i) * K + k], row(i) = order ? order[i] : i.
// A block transposes SNAP_ROWS rows through LDS so that both the row reads (K consecutive doubles) and the column
// writes (consecutive i) are coalesced.  Dynamic LDS: K * (SNAP_ROWS + 1) doubles.
// With x2 / out2 / out3 (the face fluxes): out = x, out2 = x2, out3 = x + x2 in ONE launch (round 5; three launches before) --
// `tot = adv + dif` is the same IEEE addition wherever it is made.  The outputs may be device memory (the ring's snapshot
// buffer, copied out by the copy engine) or the device alias of page-locked host memory (small snapshots: written in place).
constexpr int SNAP_ROWS = 64;
__global__ void __launch_bounds__(BLOCK) k_snapshot_t(int n_out, int K, int Kp, const int32_t* __restrict__ order,
                                                    const double* __restrict__ x, const double* __restrict__ x2, double* __restrict__ out,
                                                    double* __restrict__ out2, double* __restrict__ out3) {
  // K: the caller's constituents (what is written), Kp >= K: the row width of x (the engine's internal width)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  double* s = reinterpret_cast<double*>(s_dyn);                    // [K][SNAP_ROWS + 1]
  const int tid = threadIdx.x;
  const int nv = x2 ? 3 : 1;
  for (int i0 = blockIdx.x * SNAP_ROWS; i0 < n_out; i0 += gridDim.x * SNAP_ROWS) {
    const int rows = min(SNAP_ROWS, n_out - i0);
    for (int v = 0; v < nv; ++v) {                                 // (the second and third reading hit the cache lines of the first)
      for (int q = tid; q < rows * K; q += BLOCK) {
        const int r = q / K, k = q - r * K;
        const int src = order ? order[i0 + r] : i0 + r;
        const size_t at = (size_t)src * Kp + k;
        s[k * (SNAP_ROWS + 1) + r] = v == 0 ? x[at] : (v == 1 ? x2[at] : x[at] + x2[at]);
      }
      __syncthreads();
      double* o = v == 0 ? out : (v == 1 ? out2 : out3);
      for (int q = tid; q < rows * K; q += BLOCK) {
        const int k = q / rows, r = q - k * rows;
        o[(size_t)k * n_out + i0 + r] = s[k * (SNAP_ROWS + 1) + r];
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------ small meshes
// The reference's own meshes (Ohio River 2 943 cells, Sumwere Creek 367) fit one CU's LDS.  There a sweep launch
// is pure latency (4-6 us for a few microseconds of work), so the WHOLE Jacobi solve of one constituent runs in one
// 1024-thread workgroup: the column of x lives in LDS, every thread keeps its rows' bhat and Jacobi weights in registers;
// ||x'-x||^2 (the exact scaled residual of the sweep's input) is block-reduced every `check_every` sweeps.  One workgroup per
// constituent (columns are independent systems).
//
// Round 5 (profiles/r05_small_mesh.txt: the kernel was 317 of the step's 342 us at 2 943 cells, 2.3 us per sweep, and the time
// was the LDS gather -- 4 rows x 8 padded slots per thread where the mesh has 4.1 neighbours per row):
//   * the rows are dealt to the threads SORTED BY NEIGHBOUR COUNT (host: ensure_small_tables), so the 64 rows a wave relaxes
//     together have the same count: four neighbours are gathered without a branch, the fifth to eighth only by the waves whose
//     rows have them -- LDS reads fall from RPT x 8 to about the neighbours there are; ghost faces (weight zero in J) take no slot;
//   * the column lives in LDS in that internal order (Cuthill-McKee under the count sort), neighbours taken in ascending position:
//     the lanes of a half-wave gather from different banks (38 % of the LDS cycles were bank conflicts before);
//   * 3 rows per thread exist as a variant (2 943 cells = 2.9 x 1024);
//   * two columns in LDS (read one, write the other): ONE barrier per sweep instead of two; the thread's own previous value stays in
//     a register.
// The order of a row's sum is the ascending position of its neighbours: fixed per engine, so run to run the same bits.
constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_DEG = 8;                     // register-resident weights: real neighbours per row
// RPT  rows per thread (compile time: static register indexing).
// rows  [RPT][1024]      the row at position p = i * 1024 + thread: -1 = none
// recs  [8][RPT][1024]   the record index (into rec) of the q-th REAL neighbour of that row: -1 = none
// offs  [4][RPT][1024]   byte offsets of neighbours 2 qq (low half) and 2 qq + 1 (high half) in a column; empty: the row's own
//
// COOP (meshes of 4 097 .. ~20 000 cells, BASELINE configs 1 / 2 at "~10 k cells"): P workgroups share the rows of one constituent
// (host::build_small_plan): each holds its own rows plus `D` layers of halo rows, relaxes D sweeps on its own (the halo layers
// redundantly: the iterates are those of the global Jacobi iteration bit for bit) and then exchanges -- own rows that others
// hold as halo, and the check's partial norms -- through global memory.  The hand-off is the measured-valid form of
// MI355X_MICROARCH.md (inter-workgroup visibility, first row of the table): all payload stores and loads `sc1` (relaxed
// agent-scope atomics), every storing wave waits vmcnt(0), workgroup barrier, ONE lane adds to the constituent's arrival counter
// (agent scope), ONE lane polls it with sc1 loads, workgroup barrier, then the loads; one workgroup per CU (LDS request); two
// publication buffers alternate, so a part one exchange ahead never overwrites what a neighbour still reads.  Every spin is
// bounded: a part that waits longer than `spin_ticks` raises the abort bit of every counter and a sticky word the host reads; parts that had
// already passed that exchange leave at their next one, and the host -- whatever some of them wrote -- restores the state from its kept
// copy, takes the multi-launch path and stops using this one (see the end of the kernel).
constexpr unsigned long long SMALL_ABORT = 1ull << 62;
struct SmallCoop {
  int P, D, S, R;
  const int32_t* send_pos; const int32_t* send_cnt;       // [P][S], [P]
  const int32_t* recv_src; const int32_t* recv_pos; const int32_t* recv_cnt;   // [P][R] (part * 2 S + slot), [P][R], [P]
  double* pub;                      // [K][P][2][S]
  double* red;                      // [K][P][2][4]
  unsigned long long* arrive;       // [K], zero at launch; bit 62 = abort
  long long spin_ticks;             // bound of a wait, in wall_clock64() ticks (100 MHz)
  int fences;                       // 1: an agent-scope release in front of the arrival and an acquire behind the poll (see sync_exchange)
};
template <int RPT, bool COOP>
__global__ void __launch_bounds__(SMALL_THREADS) k_small_jacobi(
    int n, int K, const int32_t* __restrict__ rows, const int32_t* __restrict__ recs, const uint32_t* __restrict__ offs,
    const FaceRec* __restrict__ rec,
    const double* __restrict__ diag, const double* __restrict__ bhat, double* __restrict__ x, double tol2, double ew_rel, double ew_abs,
    int max_sweeps, int check_every, double* __restrict__ info /* [K][5]: sweeps, ||x'-x||^2, ||bhat||^2, max(|dx| - ew_rel |x'|), max |x'| */,
    ReduceNote note /* (round 5) the same five numbers per constituent into page-locked host memory + a sequence word: no download */,
    SmallCoop co, int first_check /* no check before this sweep (the host's guess from the last step: a check is two barriers) */) {
  extern __shared__ double s_x[];                // two columns of RPT x 1024 doubles (compile-time stride: the column a sweep
  constexpr int COL = RPT * SMALL_THREADS;       // reads or writes is an immediate offset of its LDS instructions), then scratch
  double* s_red = s_x + 2 * COL;
  const int P = COOP ? co.P : 1;
  const int k = COOP ? blockIdx.x / P : blockIdx.x;
  const int part = COOP ? blockIdx.x - k * P : 0;
  rows += (size_t)part * COL; recs += (size_t)part * SMALL_DEG * COL; offs += (size_t)part * (SMALL_DEG / 2) * COL;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  constexpr bool OWN_IN_REG = RPT <= 3;          // the row's previous value: a register, or (4 rows per thread: 128 VGPRs are short) LDS
  double bh[RPT], xo[OWN_IN_REG ? RPT : 1];
  int row[RPT], dmax[RPT];
  unsigned roff2[RPT * SMALL_DEG / 2];           // BYTE offsets of the neighbours in a column (< 32 768: two per register)
  double rw[RPT * SMALL_DEG];
  double bb = 0.0;
  unsigned ownm = 0u;                            // bit i: row slot i holds one of this part's OWN rows (counted in the norms, written back)
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int code = rows[i * SMALL_THREADS + tid];
    const int c = code < 0 ? -1 : (code & 0x0fffffff);
    const int kind = code < 0 ? 0 : (code >> 28);          // 1 own, 2 relaxed halo, 3 read-only halo
    row[i] = c; bh[i] = 0.0;
    if constexpr (OWN_IN_REG) xo[i] = 0.0;
    double rdg = 0.0;
    s_x[i * SMALL_THREADS + tid] = 0.0;           // (empty positions: bhat = 0, weights 0 -- they stay zero and need no branch)
    s_x[COL + i * SMALL_THREADS + tid] = 0.0;
    if (c >= 0) {
      const double x0 = x[(size_t)c * K + k];
      if constexpr (OWN_IN_REG) xo[i] = x0;
      s_x[i * SMALL_THREADS + tid] = x0;
      if (kind != 3) {
        bh[i] = bhat[(size_t)c * K + k];
        rdg = 1.0 / diag[c];
      }
      if (kind == 1) { bb += bh[i] * bh[i]; ownm |= 1u << i; }
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < SMALL_DEG; ++q) {
      const int j = recs[(q * RPT + i) * SMALL_THREADS + tid];
      rw[i * SMALL_DEG + q] = 0.0;                // empty slot: weight zero, the row's own offset
      if (j >= 0) {
        const FaceRec fr = rec[j];
        rw[i * SMALL_DEG + q] = (fr.d - fmin((double)fr.a_c, 0.0)) * rdg;
        cnt = q + 1;
      }
      if ((q & 1) == 0) roff2[(i * SMALL_DEG + q) / 2] = offs[((q / 2) * RPT + i) * SMALL_THREADS + tid];
    }
    if constexpr (COOP) {
      if (kind == 3) { rw[i * SMALL_DEG] = 1.0; cnt = 1; }   // a read-only halo row copies itself (0 + 1.0 * x: exact) until the exchange refreshes it
    }
    for (int off = 32; off >= 1; off >>= 1) cnt = max(cnt, __shfl_xor(cnt, off, 64));
    dmax[i] = __builtin_amdgcn_readfirstlane(cnt);            // wave-uniform: the gather loop's bound for this row slot
  }
  auto block_sum = [&](double v) -> double {      // fixed-order reduction: shuffles inside a wave, then the 16 wave sums
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double t = lane < SMALL_THREADS / 64 ? s_red[lane] : 0.0;      // (16 wave sums, folded in a fixed order on every wave alike)
    for (int off = 8; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
    return __shfl(t, 0, 64);
  };
  // the three measures of a check in ONE round (sum, max, max): two barriers instead of six
  auto block_reduce3 = [&](double& a, double& b, double& c) {
    for (int off = 32; off >= 1; off >>= 1) {
      a += __shfl_xor(a, off, 64);
      b = fmax(b, __shfl_xor(b, off, 64));
      c = fmax(c, __shfl_xor(c, off, 64));
    }
    __syncthreads();
    if (lane == 0) { s_red[wave] = a; s_red[16 + wave] = b; s_red[32 + wave] = c; }
    __syncthreads();
    if (lane < SMALL_THREADS / 64) { a = s_red[lane]; b = s_red[16 + lane]; c = s_red[32 + lane]; }
    else { a = 0.0; b = -INFINITY; c = -INFINITY; }
    for (int off = 8; off >= 1; off >>= 1) {       // (16 wave results, folded in a fixed order on every wave alike)
      a += __shfl_xor(a, off, 64);
      b = fmax(b, __shfl_xor(b, off, 64));
      c = fmax(c, __shfl_xor(c, off, 64));
    }
    a = __shfl(a, 0, 64); b = __shfl(b, 0, 64); c = __shfl(c, 0, 64);
  };
  char* const lds = reinterpret_cast<char*>(s_x);
  int dir = 0;                                   // the column holding the current iterate
  int xchg = 0;                                  // exchanges made (COOP)
  bool aborted = false;
  // COOP: this part's exchange lists, copied into LDS once (behind the reduction scratch): an exchange then has no index load from
  // global memory in front of its stores and loads (two dependent latencies of ~0.7 us each out of ~4 us per exchange)
  int32_t* const s_send = reinterpret_cast<int32_t*>(s_red + 64);
  int32_t* const s_rsrc = s_send + (COOP ? co.S : 0);
  int32_t* const s_rpos = s_rsrc + (COOP ? co.R : 0);
  if constexpr (COOP) {
    for (int i = tid; i < co.send_cnt[part]; i += SMALL_THREADS) s_send[i] = co.send_pos[(size_t)part * co.S + i];
    for (int i = tid; i < co.recv_cnt[part]; i += SMALL_THREADS) {
      s_rsrc[i] = co.recv_src[(size_t)part * co.R + i];
      s_rpos[i] = co.recv_pos[(size_t)part * co.R + i];
    }
  }                                              // (published by the barriers of block_sum(bb) below)
  // COOP: publish the own rows others hold as halo and this part's three numbers, wait for every part of the constituent, refresh
  // the halo rows of the current column, and combine the numbers (sum, max, max) in part order -- the same bits in every part.
  // Called by all threads, after a workgroup barrier that followed the column's last write (block_sum / block_reduce3 end in one).
  auto sync_exchange = [&](double& a, double& b, double& c, bool numbers = true) {   // numbers == false (uniform): the halo rows only
    if constexpr (COOP) {
      const int par = xchg & 1;
      double* col = s_x + dir * COL;
      double* mypub = co.pub + (((size_t)k * P + part) * 2 + par) * co.S;
      const int ns = co.send_cnt[part];
      for (int s2 = tid; s2 < ns; s2 += SMALL_THREADS)
        __hip_atomic_store(mypub + s2, col[s_send[s2]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 0 && numbers) {
        double* myred = co.red + (((size_t)k * P + part) * 2 + par) * 4;
        __hip_atomic_store(myred + 0, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(myred + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(myred + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave, before the barrier its signalling lane joins
      __syncthreads();
      if (tid == 0) {
        if (co.fences) {                                        // (every wave's stores are drained: ONE lane's release covers the workgroup)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (ROCm 7.2 can drop the fence's own wait: keep this one, in this order)
        }
        __hip_atomic_fetch_add(co.arrive + k, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (unsigned long long)P * (unsigned long long)(xchg + 1);
        const long long t0 = wall_clock64();
        unsigned long long seen;
        for (;;) {                                                // ONE load per poll: the abort travels in the counter's top bit
          seen = __hip_atomic_load(co.arrive + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (seen >= target) break;
          if (wall_clock64() - t0 > co.spin_ticks) {              // a part never came (not resident?): everyone leaves
            for (int kk = 0; kk < K; ++kk) __hip_atomic_fetch_or(co.arrive + kk, SMALL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen = SMALL_ABORT; break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (co.fences) {                                        // ONE acquire after the match, waited for before the barrier the loads follow
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_red[56] = seen >= SMALL_ABORT ? 1.0 : 0.0;
      }
      __syncthreads();
      if (s_red[56] != 0.0) { aborted = true; return; }
      const int nr = co.recv_cnt[part];
      const double* kpub = co.pub + (size_t)k * P * 2 * co.S + (size_t)par * co.S;
      for (int r = tid; r < nr; r += SMALL_THREADS)
        col[s_rpos[r]] = __hip_atomic_load(kpub + s_rsrc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wave == 0 && numbers) {
        double ra = 0.0, rb = -INFINITY, rc = -INFINITY;
        if (lane < P) {
          const double* r4 = co.red + (((size_t)k * P + lane) * 2 + par) * 4;
          ra = __hip_atomic_load(r4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          rb = __hip_atomic_load(r4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          rc = __hip_atomic_load(r4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int off = 8; off >= 1; off >>= 1) {                 // (P <= 16 parts, folded in a fixed order)
          ra += __shfl_xor(ra, off, 64);
          rb = fmax(rb, __shfl_xor(rb, off, 64));
          rc = fmax(rc, __shfl_xor(rc, off, 64));
        }
        if (lane == 0) { s_red[57] = ra; s_red[58] = rb; s_red[59] = rc; }
      }
      __syncthreads();                                           // the refreshed halo rows and the combined numbers, for everyone
      if (numbers) { a = s_red[57]; b = s_red[58]; c = s_red[59]; }
      ++xchg;
    }
  };
  bb = block_sum(bb);                             // (its barriers also publish the first column)
  if constexpr (COOP) { double u = -INFINITY, v = -INFINITY; sync_exchange(bb, u, v); }
  int sweep = 0, since = 0;
  double rr = 0.0, m1 = INFINITY, m2 = 0.0;      // (m1 = +inf until an element-wise verification has run)
  // one sweep: column DIR -> column 1 - DIR in LDS; the caller's barrier publishes it.  MEASURE: also the element-wise measures of
  // this sweep (taken in ONE extra verification sweep after the norm criterion holds: the hot loop stays lean)
  auto sweep_dir = [&](auto measure, auto dirc, double& dx2, double& e1, double& e2) {
    constexpr int DIR = decltype(dirc)::value;
    const char* const cur = lds + DIR * COL * 8;
    char* const nxt = lds + (1 - DIR) * COL * 8;
    auto xat = [&](int i, int q) -> double {       // (i, q compile-time after unrolling: static register indexing)
      const unsigned pr = roff2[(i * SMALL_DEG + q) / 2];
      return *reinterpret_cast<const double*>(cur + ((q & 1) ? (pr >> 16) : (pr & 0xffffu)));
    };
    // the first four neighbours of every row slot in ONE basic block -- no branch between the LDS reads, so they are all in flight
    // together (a scalar branch per neighbour made every read wait for its own latency: 1.35 us per sweep at 2 943 cells);
    // slots beyond a row's count hold weight zero and the row's own offset (empty row slots: offset 0): harmless reads
    double sum[RPT];                               // sum = (J x)[row]: x' = bhat + sum, neighbours in record order
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      sum[i] = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) sum[i] += rw[i * SMALL_DEG + q] * xat(i, q);
    }
    // neighbours five to eight: only the waves whose rows have them (the rows are dealt sorted by count: a few waves of slot 0).
    // (4 rows per thread: hipcc keeps THESE weights in scratch -- 128 VGPRs are short by that much -- and the common path free of it)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      if (dmax[i] > 4) {
        sum[i] += rw[i * SMALL_DEG + 4] * xat(i, 4);
        sum[i] += rw[i * SMALL_DEG + 5] * xat(i, 5);
        if (dmax[i] > 6) {
          sum[i] += rw[i * SMALL_DEG + 6] * xat(i, 6);
          sum[i] += rw[i * SMALL_DEG + 7] * xat(i, 7);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {                // (empty positions compute 0 = 0 + 0 and write it: no branch)
      const double xn = bh[i] + sum[i];
      const double dx = xn - (OWN_IN_REG ? xo[i] : *reinterpret_cast<const double*>(cur + (i * SMALL_THREADS + tid) * 8));
      if constexpr (COOP) dx2 += ((ownm >> i) & 1u) ? dx * dx : 0.0;     // (halo rows belong to another part's norm)
      else dx2 += dx * dx;
      if constexpr (decltype(measure)::value) {
        if ((ownm >> i) & 1u) {                    // (an empty position would put 0 - ew_rel * 0 = 0 into a maximum that may be negative)
          e1 = fmax(e1, fabs(dx) - ew_rel * fabs(xn));
          e2 = fmax(e2, fabs(xn));
        }
      }
      if constexpr (OWN_IN_REG) xo[i] = xn;
      *reinterpret_cast<double*>(nxt + (i * SMALL_THREADS + tid) * 8) = xn;
    }
  };
  auto sweep_once = [&](auto measure, double& dx2, double& e1, double& e2) {
    if (dir == 0) sweep_dir(measure, std::integral_constant<int, 0>{}, dx2, e1, e2);
    else sweep_dir(measure, std::integral_constant<int, 1>{}, dx2, e1, e2);
    dir ^= 1;
  };
  for (; !aborted;) {
    double dx2 = 0.0, e1 = -INFINITY, e2 = 0.0;
    sweep_once(std::false_type{}, dx2, e1, e2);
    ++sweep; ++since;
    // COOP: the halo layers carry co.D sweeps; the check rides on the exchange
    // (check_every is 4 wherever the kernel is launched: a mask, not a division by a run-time value every sweep)
    const bool check = (COOP ? since >= co.D : (sweep >= first_check && (check_every == 4 ? (sweep & 3) == 0 : sweep % check_every == 0))) || sweep >= max_sweeps;
    if (check) {                                 // uniform
      if constexpr (COOP) {
        if (sweep < first_check && sweep < max_sweeps) {        // too early for the norm to matter (the host's guess): the halo rows only
          __syncthreads();                       // (the column's last writes, in front of the exchange's reads)
          double u = 0.0, v = 0.0, w = 0.0;
          sync_exchange(u, v, w, false); since = 0;
          if (aborted) break;
          continue;
        }
      }
      rr = block_sum(dx2);                       // (its barriers also publish the new column)
      if constexpr (COOP) { double u = -INFINITY, v = -INFINITY; sync_exchange(rr, u, v); since = 0; if (aborted) break; }
      if (!(rr == rr) || sweep >= max_sweeps) break;          // NaN, or out of sweeps
      if (!(rr > tol2 * bb)) {                   // the norm criterion holds: verify the element-wise rule with one more sweep
        dx2 = 0.0;
        sweep_once(std::true_type{}, dx2, e1, e2);
        ++sweep;
        block_reduce3(dx2, e1, e2);
        if constexpr (COOP) { sync_exchange(dx2, e1, e2); since = 0; if (aborted) break; }
        rr = dx2; m1 = e1; m2 = e2;              // ||x'-x||^2 and the element-wise measures of this sweep (k_apply MODE 4)
        if (!(rr == rr) || (!(rr > tol2 * bb) && !(m1 > ew_abs * m2)) || sweep >= max_sweeps) break;
      }
    } else {
      __syncthreads();
    }
  }
  // (round 6) The parts do NOT agree on an abort by themselves: a part may time out while the last arriver completes the
  // target -- the others then pass that exchange, and if it was the final one they write their rows while the part that gave up
  // leaves its own untouched.  So every part that aborted says so where the host looks -- a sticky word behind the K x 5 numbers
  // (device copy and page-locked copy) -- EVERY workgroup, not only the K that carry numbers, arrives at the notification counter
  // before the sequence word is published, and the host restores the state from its kept copy whenever the word is set: what the
  // parts that passed may have written never counts.
  if (aborted) {
    if (tid == 0 && part == 0) { sweep = -1; rr = 0.0; }
  } else {
#pragma unroll
    for (int i = 0; i < RPT; ++i)
      if ((ownm >> i) & 1u) x[(size_t)row[i] * K + k] = OWN_IN_REG ? xo[i] : s_x[dir * COL + i * SMALL_THREADS + tid];
  }
  if (tid == 0) {
    if (part == 0) { info[k * 5 + 0] = (double)sweep; info[k * 5 + 1] = rr; info[k * 5 + 2] = bb; info[k * 5 + 3] = m1; info[k * 5 + 4] = m2; }
    if constexpr (COOP) {
      if (aborted) {
        __hip_atomic_store(info + (size_t)K * 5, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (note.host_seq) __hip_atomic_store(note.host_out + (size_t)K * 5 + 1, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (note.host_seq) {
      if (part == 0) {
        note.host_out[k * 5 + 0] = (double)sweep; note.host_out[k * 5 + 1] = rr; note.host_out[k * 5 + 2] = bb; note.host_out[k * 5 + 3] = m1; note.host_out[k * 5 + 4] = m2;
      }
      __threadfence_system();
      const unsigned prev = atomicAdd(note.arrive, 1u);
      if (prev == (unsigned)(K * P) - 1u) {        // the last workgroup of the launch: every number and every abort word is out
        __threadfence_system();
        *note.arrive = 0u;
        const unsigned long long sq = *note.dev_seq + 1ull;
        *note.dev_seq = sq;
        __hip_atomic_store(note.host_seq, sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ a-8 on device
// Device-resident reaction step between two transport steps: c[cell, :] <- M c[cell, :] for every owned cell, M a
// K x K matrix (first-order decay on the diagonal, pairwise exchange off it) -- the in-HBM stand-in for the host
// callback of transport.py:233-236 (update_concentration), without the D2H / H2D round trip of the state.
__global__ void __launch_bounds__(BLOCK) k_react_linear(int n_rows, int K, const double* __restrict__ M, double* __restrict__ c) {
  extern __shared__ double s_tile[];               // [rows_per_block][K] state rows, then M
  const int rows_pb = BLOCK / K;
  double* s_M = s_tile + rows_pb * K;
  for (int i = threadIdx.x; i < K * K; i += BLOCK) s_M[i] = M[i];
  const int r = threadIdx.x / K, k = threadIdx.x - r * K;
  for (int base = blockIdx.x * rows_pb; base < n_rows; base += gridDim.x * rows_pb) {
    const int cell = base + r;
    const bool live = (r < rows_pb) && (cell < n_rows);
    __syncthreads();
    if (live) s_tile[r * K + k] = c[(size_t)cell * K + k];
    __syncthreads();
    if (live) {
      double acc = 0.0;
      for (int j = 0; j < K; ++j) acc += s_M[k * K + j] * s_tile[r * K + j];
      c[(size_t)cell * K + k] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------ halo
__global__ void __launch_bounds__(BLOCK) k_pack_rows(int64_t total, int K, const int32_t* __restrict__ cells,
                                                   const double* __restrict__ vec, double* __restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= total) return;
  const int64_t row = i / K;
  const int k = (int)(i - row * K);
  buf[i] = vec[(size_t)cells[row] * K + k];
}

// vec2 (optional): the ping-pong partner of vec receives the same rows
__global__ void __launch_bounds__(BLOCK) k_unpack_rows(int64_t total, int K, const int32_t* __restrict__ cells,
                                                     const double* __restrict__ buf, double* __restrict__ vec,
                                                     double* __restrict__ vec2) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= total) return;
  const int64_t row = i / K;
  const int k = (int)(i - row * K);
  const size_t o = (size_t)cells[row] * K + k;
  const double v = buf[i];
  vec[o] = v;
  if (vec2) vec2[o] = v;
}

}  // namespace cwr
