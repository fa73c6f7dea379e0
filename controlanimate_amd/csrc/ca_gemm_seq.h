// Sequence-number table for LDS-DMA completion flags (ca_gemm_ps.h, ca_gemm_wres.h).
//
// On gfx950 loads and stores share the VMEM counter and retire out of order with respect to each other: a wave that streams
// operands by LDS-DMA AND stores results can only wait for "all but the N youngest loads" by also waiting for its stores.
// Loads return IN ORDER among themselves, so a 4-byte LDS-DMA issued after a unit's data pieces, fetching the unit's
// sequence number from this table into an LDS flag word, lands after them: the wave polls that word with a ds_read instead
// of waiting on the counter (tools/probe_flag.hip).  table[i] = i; a flag slot is re-used with a DIFFERENT number.
#pragma once

struct CaSeqTable {
  unsigned v[1024];
};
constexpr CaSeqTable ca_make_seq_table() {
  CaSeqTable t{};
  for (int i = 0; i < 1024; ++i) t.v[i] = (unsigned)i;
  return t;
}
__device__ const CaSeqTable ca_seq_table = ca_make_seq_table();
