// chain_sync.h -- overlapped decode chain: consecutive launches of a decode step alternate between two streams and carry
// their dependency in a counter in device memory instead of the kernel boundary (DESIGN.md section 3a'').
//
// Why: inside one stream launch N+1 starts when launch N has drained; its fixed start-up (kernel-argument fetch, scale
// tables, filling the weight ring: ~4 us, none of which depends on launch N) is then serial.  On the other stream it runs
// as soon as workgroups of launch N leave their CUs; it waits for N's signals only before it reads activations.
//
// Progress: launch k sits behind launch k-2 in its own (in-order) stream, so at most two launches are in flight; the older
// one never waits on the younger, and the younger's workgroups only spin while holding CUs the older no longer needs
// (its workgroups were all placed before k-2 ... k-1 finished).  The spin is bounded (hw.h: FLAG_SPIN_LIMIT).
// Start of the chain: launches 0 and 1 would be released together and share the CUs between them -- launch 1 spinning on
// CUs that launch 0 still needs.  So every workgroup of launch 0 reports on entry, and a one-wave gate kernel ahead of
// launch 1 in the other stream lets it through only when all of launch 0 is resident.  (The first launch must be a q_gemm.)
//
// While a chain is open (exl2_chain_overlap_begin ... _end) every chained launch of THIS thread -- the chained q_gemm
// entry points and exl2_attn_decode_fused -- takes its stream, its wait counter / target and its signal counter from here.
#pragma once
#include "hw.h"


struct ChainLaunch { const u32* wait; u32* signal; u32* arrive; void* stream; };      // blocks of SYNC_BLOCK_WORDS (hw.h)

// true while a chain is open on this thread
bool chain_sync_active();
// stream / counters of the next launch; returns < 0 (error set) when the chain has run out of counters
int chain_sync_next(ChainLaunch* out);
// the launch took place; `arrivals` = its workgroups (first launch only: the gate on the other stream is launched with this
// target).  Returns < 0 on error.
int chain_sync_done(u32 arrivals);
