// K3 EKF update + log-weight: the general kernel (supplied ids, or per-particle ids from the
// association kernel) and the loop-free single-sighting kernel.  The maximum-likelihood variants
// that settle the association themselves are in pk_k_observe_ml.hip.
//
// Hand-written gfx950 (CDNA4, wave64) kernels of the FastSLAM particle update; see DESIGN.md
// section 4.  No MFMA: the algebra is 2x2 / 3x3 and register resident (pk_math.hpp).
#include "pk_device.hpp"

namespace pk {

// The updated map rows are written with non-temporal stores: nothing reads them before the next
// step, and keeping them out of L2 / Infinity Cache leaves the caches to the read stream (the
// duplicates a resample leaves read shared source slots): the supplied-ids kernel went from 0.190
// to 0.171 ms at 10 000 x 500 with nothing else changed.
typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ K3 observe (EKF + weight)
struct ObserveArgs {
  SlotSource ss;
  unsigned char* map_dst;
  size_t slot_bytes, count_off;
  int32_t* src;  // in: slot of particle p in map_src; out: identity
  const double *x, *y;
  double* logw;
  const double* blobs;   // B x 4
  const double* blobdir; // ML: B x 2 unit ray directions (closest_point :510)
  const int32_t* first;  // KNOWN: [L] first blob matched to landmark l, or -1
  const int32_t* next;   // KNOWN: [B] next blob matched to the same landmark, or -1
  int32_t* ids;          // ML: [P x B]; a tentative id whose probability is 0 is reset to 0
  const unsigned char* immutable;
  int n_unmatched;  // KNOWN: blobs with id 0
  const unsigned char* only_flagged;  // when set: skip particles whose flag is 0 (k_observe_fast did them)
  const unsigned* n_flagged;          // with only_flagged: number of flagged particles (0 -> nothing to do)
  int reset;                          // 1: the weight restarts from 1 (prkt_core_v2.py:73) instead of accumulating
  unsigned long long* gmax_key;       // running max of the new log-weights (double_to_key), or NULL
  int L, Lp, B;
  int64_t P;
  Noise<double> qt;
};

// Is probability_of_match(...) > 0 for a pair that already passed both gates (:433, :441)?
// The association kernel leaves this to us for blobs with a single gate-passing landmark,
// because the landmark's covariance is in registers here.  pr = (500 exp(a1)) (500 exp(a2))
// / 250000 with a1, a2 the two log-pdfs; whenever a1 + a2 is far from the float64 underflow
// edge the answer is known without evaluating a single exp/log; otherwise evaluate it
// exactly as the reference does.  pse = atan2(f.my - sy, f.mx - sx).
__device__ __forceinline__ bool match_is_positive(const Landmark<double>& f, double sx, double sy, double pse,
                                                  const BlobT<double>& z, double ux, double uy) {
  if (fabs(pse - z.bearing) > Consts<double>::half_pi) return false;  // :473-475 -> bp = 0
  double nx, ny;
  closest_point(f.mx, f.my, sx, sy, ux, uy, nx, ny);
  const double ex = nx - f.mx, ey = ny - f.my;
  const double det2 = f.pxx * f.pyy - f.pxy * f.pxy;
  const double maha2 = (f.pyy * ex * ex - 2.0 * f.pxy * ex * ey + f.pxx * ey * ey) / det2;
  double det3;
  const Sym3<double> inv = sym3_inverse(Sym3<double>{f.crr, f.crg, f.crb, f.cgg, f.cgb, f.cbb}, det3);
  const double maha3 = sym3_quad(inv, z.r - f.mr, z.g - f.mg, z.b - f.mb);
  // log det <= 138.2 for det <= 1e60, so a1 + a2 >= -0.5 (9.2 + 276.4 + 800) > -543: no underflow
  if (det2 > 0.0 && det2 < 1e60 && det3 > 0.0 && det3 < 1e60 && maha2 >= 0.0 && maha3 >= 0.0 &&
      maha2 + maha3 < 800.0)
    return true;
  const double bp = 500.0 * exp(-0.5 * (2.0 * Consts<double>::log_two_pi + log(det2) + maha2));
  const double cp = 500.0 * exp(-0.5 * (3.0 * Consts<double>::log_two_pi + log(det3) + maha3));
  return bp * cp / 250000.0 > 0.0;
}

__device__ __forceinline__ BlobT<double> load_blob(const double* blobs, int b) {
  const double2 z01 = *reinterpret_cast<const double2*>(blobs + 4 * (size_t)b);
  const double2 z23 = *reinterpret_cast<const double2*>(blobs + 4 * (size_t)b + 2);
  return BlobT<double>{z01.x, z01.y, z23.x, z23.y};
}

// All blobs matched to landmark l, in scan order (prkt_core_v2.py:88).
// ML: the ids are tentative.  Association saw the state BEFORE any update (:84), so first
// settle every blob of the chain against the untouched state (s_ids[b] = 0 drops it), then
// apply the surviving ones sequentially.
template <bool KNOWN>
__device__ __forceinline__ double apply_blobs(Landmark<double>& lm, int l, double sx, double sy,
                                              const ObserveArgs& a, const int32_t* first,
                                              const int32_t* next, int32_t* s_ids, int32_t* gid) {
  double acc = 0.0;
  const int b0 = first[l];
  if (b0 < 0) return acc;
  const bool imm = a.immutable[l] != 0;
  const double pse = pk_atan2(lm.my - sy, lm.mx - sx);
  if (!KNOWN) {
    for (int b = b0; b >= 0; b = next[b]) {
      const BlobT<double> z = load_blob(a.blobs, b);
      const double2 dir = *reinterpret_cast<const double2*>(a.blobdir + 2 * (size_t)b);
      if (!match_is_positive(lm, sx, sy, pse, z, dir.x, dir.y)) {
        s_ids[b] = 0;
        gid[b] = 0;
      }
    }
  }
  bool fresh = true;  // lm still equals the state pse was computed from
  for (int b = b0; b >= 0; b = next[b]) {
    if (!KNOWN && s_ids[b] == 0) {
      acc += Consts<double>::log_no_match;  // unseen feature: weight *= 0.1 (:94-95)
      continue;
    }
    const BlobT<double> z = load_blob(a.blobs, b);
    acc += ekf_update(lm, sx, sy, z, a.qt, imm, (EkfAux<double>*)nullptr, fresh ? &pse : (const double*)nullptr);
    fresh = imm;
  }
  return acc;
}

template <bool KNOWN, int NV>
__global__ void __launch_bounds__(kObsThreads) k_observe(ObserveArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double red[kObsThreads / kWave];
  if (a.only_flagged && *a.n_flagged == 0u) return;  // workgroup-uniform: nobody was handed on
  // One workgroup per particle -- or, behind a one-pass kernel (only_flagged), a few thousand workgroups that walk the flags: the
  // stand-by launch of every step then costs 2 us instead of the 22 us that 100 000 workgroups returning at once take to dispatch.
  for (int64_t p = blockIdx.x; p < a.P; p += gridDim.x) {
  if (a.only_flagged && a.only_flagged[p] != 1) continue;  // workgroup-uniform
  const int tid = threadIdx.x;
  const int32_t sp = a.src[p];
  const unsigned char* sslot = a.ss.at(sp);
  unsigned char* dslot = a.map_dst + (size_t)p * a.slot_bytes;
  const double* sf = reinterpret_cast<const double*>(sslot);
  double* df = reinterpret_cast<double*>(dslot);
  const int* sc = reinterpret_cast<const int*>(sslot + a.count_off);
  int* dc = reinterpret_cast<int*>(dslot + a.count_off);
  const double sx = a.x[p], sy = a.y[p];
  const int Lp = a.Lp;

  const int32_t* first = a.first;
  const int32_t* next = a.next;
  int n_unmatched = a.n_unmatched;
  int32_t* gid_mut = KNOWN ? nullptr : a.ids + (size_t)p * a.B;
  int32_t* s_ids_mut = nullptr;
  if (!KNOWN) {
    // Build the per-particle landmark -> blob chains in LDS from this particle's ids.
    // Blobs are applied in scan order (prkt_core_v2.py:88): first[l] is the lowest blob
    // index matched to l, next[b] the following blob matched to the same landmark.
    int32_t* s_first = reinterpret_cast<int32_t*>(smem);
    int32_t* s_next = s_first + Lp;
    int32_t* s_ids = s_next + a.B;
    s_ids_mut = s_ids;
    const int32_t* gid = gid_mut;
    for (int l = tid; l < Lp; l += blockDim.x) s_first[l] = INT_MAX;
    for (int b = tid; b < a.B; b += blockDim.x) {
      s_ids[b] = gid[b];
      s_next[b] = -1;
    }
    __syncthreads();
    int cnt0 = 0;
    for (int b = tid; b < a.B; b += blockDim.x) {
      int id = s_ids[b];
      if (id > 0)
        atomicMin(&s_first[id - 1], b);
      else
        ++cnt0;
    }
    __syncthreads();
    for (int b = tid; b < a.B; b += blockDim.x) {
      int id = s_ids[b];
      if (id > 0 && s_first[id - 1] != b) {  // not the first sighting: link from my predecessor
        int q = b - 1;
        while (s_ids[q] != id) --q;  // terminates: first[id-1] < b has this id
        s_next[q] = b;
      }
    }
    for (int l = tid; l < Lp; l += blockDim.x)
      if (s_first[l] == INT_MAX) s_first[l] = -1;
    // number of unmatched blobs of this particle (weight *= 0.1 each, :94-95)
    double c = block_sum<kObsThreads / kWave>((double)cnt0, red);
    n_unmatched = (int)c;
    __syncthreads();
    first = s_first;
    next = s_next;
  }

  double acc = 0.0;
  if (NV == 2) {
    // two adjacent landmarks per lane: 16-byte loads/stores, 14 rows x 1 KiB per wave instruction
    for (int l0 = 2 * tid; l0 < Lp; l0 += 2 * kObsThreads) {
      double2 v[F_COUNT_FIELDS];
#pragma unroll
      for (int f = 0; f < F_COUNT_FIELDS; ++f) v[f] = *reinterpret_cast<const double2*>(sf + (size_t)f * Lp + l0);
      int2 c = *reinterpret_cast<const int2*>(sc + l0);
      Landmark<double> A{v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x,
                         v[8].x, v[9].x, v[10].x, v[11].x, v[12].x, v[13].x, c.x};
      Landmark<double> Bq{v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y,
                          v[8].y, v[9].y, v[10].y, v[11].y, v[12].y, v[13].y, c.y};
      if (l0 < a.L) acc += apply_blobs<KNOWN>(A, l0, sx, sy, a, first, next, s_ids_mut, gid_mut);
      if (l0 + 1 < a.L) acc += apply_blobs<KNOWN>(Bq, l0 + 1, sx, sy, a, first, next, s_ids_mut, gid_mut);
      __builtin_nontemporal_store(d2v{A.mx, Bq.mx}, reinterpret_cast<d2v*>(df + (size_t)F_MX * Lp + l0));
      __builtin_nontemporal_store(d2v{A.my, Bq.my}, reinterpret_cast<d2v*>(df + (size_t)F_MY * Lp + l0));
      __builtin_nontemporal_store(d2v{A.mr, Bq.mr}, reinterpret_cast<d2v*>(df + (size_t)F_MR * Lp + l0));
      __builtin_nontemporal_store(d2v{A.mg, Bq.mg}, reinterpret_cast<d2v*>(df + (size_t)F_MG * Lp + l0));
      __builtin_nontemporal_store(d2v{A.mb, Bq.mb}, reinterpret_cast<d2v*>(df + (size_t)F_MB * Lp + l0));
      __builtin_nontemporal_store(d2v{A.pxx, Bq.pxx}, reinterpret_cast<d2v*>(df + (size_t)F_PXX * Lp + l0));
      __builtin_nontemporal_store(d2v{A.pxy, Bq.pxy}, reinterpret_cast<d2v*>(df + (size_t)F_PXY * Lp + l0));
      __builtin_nontemporal_store(d2v{A.pyy, Bq.pyy}, reinterpret_cast<d2v*>(df + (size_t)F_PYY * Lp + l0));
      __builtin_nontemporal_store(d2v{A.crr, Bq.crr}, reinterpret_cast<d2v*>(df + (size_t)F_CRR * Lp + l0));
      __builtin_nontemporal_store(d2v{A.crg, Bq.crg}, reinterpret_cast<d2v*>(df + (size_t)F_CRG * Lp + l0));
      __builtin_nontemporal_store(d2v{A.crb, Bq.crb}, reinterpret_cast<d2v*>(df + (size_t)F_CRB * Lp + l0));
      __builtin_nontemporal_store(d2v{A.cgg, Bq.cgg}, reinterpret_cast<d2v*>(df + (size_t)F_CGG * Lp + l0));
      __builtin_nontemporal_store(d2v{A.cgb, Bq.cgb}, reinterpret_cast<d2v*>(df + (size_t)F_CGB * Lp + l0));
      __builtin_nontemporal_store(d2v{A.cbb, Bq.cbb}, reinterpret_cast<d2v*>(df + (size_t)F_CBB * Lp + l0));
      __builtin_nontemporal_store(i2v{A.count, Bq.count}, reinterpret_cast<i2v*>(dc + l0));
    }
  } else {
    // one landmark per lane: half the registers, twice the waves in flight
    for (int l = tid; l < Lp; l += kObsThreads) {
      Landmark<double> A = load_landmark(sf, sc, Lp, l);
      if (l < a.L) acc += apply_blobs<KNOWN>(A, l, sx, sy, a, first, next, s_ids_mut, gid_mut);
      __builtin_nontemporal_store(A.mx, &df[(size_t)F_MX * Lp + l]);
      __builtin_nontemporal_store(A.my, &df[(size_t)F_MY * Lp + l]);
      __builtin_nontemporal_store(A.mr, &df[(size_t)F_MR * Lp + l]);
      __builtin_nontemporal_store(A.mg, &df[(size_t)F_MG * Lp + l]);
      __builtin_nontemporal_store(A.mb, &df[(size_t)F_MB * Lp + l]);
      __builtin_nontemporal_store(A.pxx, &df[(size_t)F_PXX * Lp + l]);
      __builtin_nontemporal_store(A.pxy, &df[(size_t)F_PXY * Lp + l]);
      __builtin_nontemporal_store(A.pyy, &df[(size_t)F_PYY * Lp + l]);
      __builtin_nontemporal_store(A.crr, &df[(size_t)F_CRR * Lp + l]);
      __builtin_nontemporal_store(A.crg, &df[(size_t)F_CRG * Lp + l]);
      __builtin_nontemporal_store(A.crb, &df[(size_t)F_CRB * Lp + l]);
      __builtin_nontemporal_store(A.cgg, &df[(size_t)F_CGG * Lp + l]);
      __builtin_nontemporal_store(A.cgb, &df[(size_t)F_CGB * Lp + l]);
      __builtin_nontemporal_store(A.cbb, &df[(size_t)F_CBB * Lp + l]);
      __builtin_nontemporal_store(A.count, &dc[l]);
    }
  }
  double tot = block_sum<kObsThreads / kWave>(acc, red);
  if (tid == 0) {
    const double v = (a.reset ? 0.0 : a.logw[p]) + tot + (double)n_unmatched * Consts<double>::log_no_match;
    a.logw[p] = v;
    if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));  // sharded: same-address atomics serialise
    a.src[p] = (int32_t)p;
  }
  __syncthreads();  // (the LDS tables and the reduction's scratch are the next particle's)
  }
}


// ------------------------------------------------------------------ K3 (known ids, one sighting per landmark)
// The common supplied-ids scan on a map of <= 1024 landmarks: no landmark is matched by more than
// one blob.  One landmark per lane and no loop at all: straight-line code that the compiler fits
// in 96 VGPRs (the looped k_observe needs 163-194), so 16-20 waves per CU stream the map instead
// of 8-12.  One workgroup of THREADS >= Lp lanes per particle.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_observe_single(ObserveArgs a) {
  __shared__ double red[THREADS / kWave];
  const int64_t p = blockIdx.x;
  const int l = threadIdx.x;
  const unsigned char* sslot = a.ss.at(a.src[p]);
  unsigned char* dslot = a.map_dst + (size_t)p * a.slot_bytes;
  const double* sf = reinterpret_cast<const double*>(sslot);
  double* df = reinterpret_cast<double*>(dslot);
  const int* sc = reinterpret_cast<const int*>(sslot + a.count_off);
  int* dc = reinterpret_cast<int*>(dslot + a.count_off);
  const double sx = a.x[p], sy = a.y[p];
  const int Lp = a.Lp;
  double acc = 0.0;
  if (l < Lp) {
    Landmark<double> A = load_landmark(sf, sc, Lp, l);
    if (l < a.L) {
      const int b = a.first[l];
      if (b >= 0) {
        const BlobT<double> z = load_blob(a.blobs, b);
        acc = ekf_update(A, sx, sy, z, a.qt, a.immutable[l] != 0);
      }
    }
    __builtin_nontemporal_store(A.mx, &df[(size_t)F_MX * Lp + l]);
    __builtin_nontemporal_store(A.my, &df[(size_t)F_MY * Lp + l]);
    __builtin_nontemporal_store(A.mr, &df[(size_t)F_MR * Lp + l]);
    __builtin_nontemporal_store(A.mg, &df[(size_t)F_MG * Lp + l]);
    __builtin_nontemporal_store(A.mb, &df[(size_t)F_MB * Lp + l]);
    __builtin_nontemporal_store(A.pxx, &df[(size_t)F_PXX * Lp + l]);
    __builtin_nontemporal_store(A.pxy, &df[(size_t)F_PXY * Lp + l]);
    __builtin_nontemporal_store(A.pyy, &df[(size_t)F_PYY * Lp + l]);
    __builtin_nontemporal_store(A.crr, &df[(size_t)F_CRR * Lp + l]);
    __builtin_nontemporal_store(A.crg, &df[(size_t)F_CRG * Lp + l]);
    __builtin_nontemporal_store(A.crb, &df[(size_t)F_CRB * Lp + l]);
    __builtin_nontemporal_store(A.cgg, &df[(size_t)F_CGG * Lp + l]);
    __builtin_nontemporal_store(A.cgb, &df[(size_t)F_CGB * Lp + l]);
    __builtin_nontemporal_store(A.cbb, &df[(size_t)F_CBB * Lp + l]);
    __builtin_nontemporal_store(A.count, &dc[l]);
  }
  const double tot = block_sum<THREADS / kWave>(acc, red);
  if (l == 0) {
    const double v = (a.reset ? 0.0 : a.logw[p]) + tot + (double)a.n_unmatched * Consts<double>::log_no_match;
    a.logw[p] = v;
    if (a.gmax_key) atomicMax(a.gmax_key + (p & (kGmaxKeys - 1)), double_to_key(v));
    a.src[p] = (int32_t)p;
  }
}

int g_observe_nv = 0;  // tuning: 0 = default per variant, 1 / 2 = landmarks per lane

void launch_observe(hipStream_t s, DeviceState& d, const double* blobs_dev, const double* blobdir_dev, int B,
                    const int32_t* first_dev, const int32_t* next_dev, int n_unmatched, int32_t* ids_dev,
                    const NoiseD& qt, const ObserveExtras& ex) {
  if (d.P == 0) return;
  ObserveArgs a;
  a.ss = slot_source(d);
  a.map_dst = d.map[d.mcur ^ 1];
  a.slot_bytes = d.lay.slot_bytes;
  a.count_off = d.lay.count_off;
  a.src = d.src[d.cur];
  a.x = d.x[d.cur];
  a.y = d.y[d.cur];
  a.logw = d.logw[d.cur];
  a.blobs = blobs_dev;
  a.blobdir = blobdir_dev;
  a.first = first_dev;
  a.next = next_dev;
  a.ids = ids_dev;
  a.immutable = d.immutable;
  a.n_unmatched = n_unmatched;
  a.only_flagged = ex.only_flagged;
  a.n_flagged = ex.n_flagged;
  a.reset = ex.reset ? 1 : 0;
  a.gmax_key = ex.gmax_key;
  a.L = d.lay.L;
  a.Lp = d.lay.Lp;
  a.B = B;
  a.P = d.P;
  a.qt = make_noise(qt.q00, qt.rr, qt.rg, qt.rb, qt.gg, qt.gb, qt.bb);
  // (behind a one-pass kernel few particles, if any, are flagged: 4 096 workgroups walk the flags)
  const unsigned grid = (unsigned)(ex.only_flagged && d.P > 4096 ? 4096 : d.P);
  if (ids_dev == nullptr && ex.single_sightings && !ex.only_flagged && d.lay.Lp <= 1024 && g_observe_nv == 0) {
    if (d.lay.Lp <= 256)
      hipLaunchKernelGGL((k_observe_single<256>), dim3((unsigned)d.P), dim3(256), 0, s, a);
    else if (d.lay.Lp <= 512)
      hipLaunchKernelGGL((k_observe_single<512>), dim3((unsigned)d.P), dim3(512), 0, s, a);
    else
      hipLaunchKernelGGL((k_observe_single<1024>), dim3((unsigned)d.P), dim3(1024), 0, s, a);
  } else if (ids_dev == nullptr) {
    if (g_observe_nv == 1)
      hipLaunchKernelGGL((k_observe<true, 1>), dim3(grid), dim3(kObsThreads), 0, s, a);
    else
      hipLaunchKernelGGL((k_observe<true, 2>), dim3(grid), dim3(kObsThreads), 0, s, a);
  } else {
    const size_t lds = observe_general_lds_bytes(d.lay.Lp, B);  // <= kMaxDynLds: checked by the caller
    static bool attr_set[kMaxDevices] = {false};
    if (first_time_on_this_device(attr_set)) {
      for (const void* fn : {reinterpret_cast<const void*>(k_observe<false, 1>), reinterpret_cast<const void*>(k_observe<false, 2>)})
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds) != hipSuccess)
          (void)hipGetLastError();
    }
    if (g_observe_nv == 2)
      hipLaunchKernelGGL((k_observe<false, 2>), dim3(grid), dim3(kObsThreads), lds, s, a);
    else
      hipLaunchKernelGGL((k_observe<false, 1>), dim3(grid), dim3(kObsThreads), lds, s, a);
  }
  if (ex.flip) {
    d.mcur ^= 1;
    d.alt = nullptr;  // every slot was rewritten into the particle's own map buffer
  }
}

}  // namespace pk
