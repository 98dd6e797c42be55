// What the publish / subscribe kernels (pk_k_step_pub.hip) and the per-scan kernel that lays their table out (pk_k_cand_entries.hip)
// agree on.  Hand-written gfx950 (CDNA4, wave64).
#pragma once

namespace pk {

constexpr int kPubThreads = 512;        // the large instances' workgroup
constexpr int kPubSmallThreads = 256;   // ... the L <= 512 instance's
constexpr int kPubTailWords = 256;  // words behind glist[B]: the octet orders of k_step_pub (128 u16) and k_step_pub_big (384 u16); then rbase[16]
constexpr int kPubBigPlaces = 384;  // k_step_pub_big: six chunks of 64 octets (kPubBigMaxL / 16)
constexpr int kPubOctets = 64;  // groups of eight lanes in a 512-lane workgroup: sixteen adjacent landmarks per pair each
constexpr int kPubSlots = 4;  // gate-passing blobs a landmark keeps; more: the particle is flagged

}  // namespace pk
