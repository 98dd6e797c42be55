// HBM layout of the particle state.
//
// Poses are struct-of-arrays over particles: x[P], y[P], h[P], logw[P] plus the map
// indirection src[P] (which map slot a particle's landmarks currently live in).
//
// Maps: one contiguous *slot* per particle,
//     slot = [ 14 fields ][ Lp ] of T   followed by   [ Lp ] of int32 update counts
// with the landmark index fastest, so that a workgroup that owns one particle streams
// 14 perfectly coalesced rows.  Lp = L rounded up to a multiple of 16, so that every row starts
// on a 128-byte line and a wave's 512-byte access covers exactly four lines (with Lp = L = 500 a
// wave straddled five, and neighbouring waves shared the boundary lines: the supplied-ids kernel
// runs 0.197 ms instead of 0.213 ms at 10 000 x 500 with the 2.4 % of padding); slot_bytes is
// rounded up to 256 B.
#pragma once
#include <cstddef>
#include <cstdint>

namespace pk {

enum Field : int {
  F_MX = 0, F_MY, F_MR, F_MG, F_MB,
  F_PXX, F_PXY, F_PYY,
  F_CRR, F_CRG, F_CRB, F_CGG, F_CGB, F_CBB,
  F_COUNT_FIELDS = 14
};

// The DENSE layout (maps whose covariances couple position and colour, or are not symmetric: what the reference
// accepts, prkt_core_v2.py:882-895, but its own update never produces from block-diagonal inputs) keeps the
// reference's full state: rows 0-4 the mean, rows 5 + 5 i + j the covariance entry [i][j] -- 30 rows, 240 B per
// landmark.  Only the slow, general dense kernels (pk_k_dense.hip) read it.
constexpr int kDenseFields = 30;

struct MapLayout {
  int L;              // landmarks per particle
  int Lp;             // padded to a multiple of 16 (rows on 128-byte lines; even, for 16-byte loads of two landmarks)
  size_t slot_bytes;  // multiple of 256
  size_t count_off;   // byte offset of the int32 counts inside a slot
  int fields;         // rows of T per slot: F_COUNT_FIELDS (compact) or kDenseFields

  static MapLayout make(int L, size_t scalar, int fields = F_COUNT_FIELDS) {
    MapLayout m;
    m.L = L;
    m.Lp = (L + 15) & ~15;
    if (m.Lp == 0) m.Lp = 16;
    m.fields = fields;
    m.count_off = (size_t)fields * m.Lp * scalar;
    size_t raw = m.count_off + (size_t)m.Lp * sizeof(int32_t);
    m.slot_bytes = (raw + 255) & ~(size_t)255;
    return m;
  }
};

}  // namespace pk
