// chain_sync.h -- overlapped decode chain: consecutive launches of a decode step alternate between two streams and carry
// their dependency in a counter in device memory instead of the kernel boundary (DESIGN.md section 3a'').
//
// Why: inside one stream launch N+1 starts when launch N has drained; its fixed start-up (kernel-argument fetch, scale
// tables, filling the weight ring: ~4 us, none of which depends on launch N) is then serial.  On the other stream it runs
// as soon as workgroups of launch N leave their CUs; it waits for N's signals only before it reads activations.
//
// Progress: launch k sits behind launch k-2 in its own (in-order) stream, so at most two launches are in flight; the older
// one never waits on the younger.  The younger's workgroups spin while they hold their slots, so they must never hold a
// slot the older still needs: every workgroup of every launch reports on entry (hw.h: sync_report_entry), and a one-wave
// GATE kernel sits ahead of launch k + 1 in its stream that lets it through only when all workgroups of launch k have
// entered (round 2 had such a gate for the first pair only and met the race it leaves: about once in 150 steps a launch
// two ahead took the slots of an attention launch that was still waiting to be placed).  The spins are bounded
// (hw.h: FLAG_SPIN_LIMIT): a protocol error shows as wrong numbers and a count in the block, never as a hung GPU.
//
// While a chain is open (exl2_chain_overlap_begin ... _end) every chained launch of THIS thread -- the chained q_gemm
// entry points and exl2_attn_decode_fused -- takes its stream, its wait counter / target and its signal counter from here.
#pragma once
#include "hw.h"


struct ChainLaunch { const u32* wait; u32* signal; u32* arrive; void* stream; };      // blocks of SYNC_BLOCK_WORDS (hw.h)

// true while a chain is open on this thread
bool chain_sync_active();
// 5 .. 16 rows (round 4): the chain's (xp, ss) hand-off buffers in the MFMA-tiled layout (qgemv_flat.h: FlatIn.a_tiled) between
// exl2_chain_set_tiled(1) and (0): every producer of the residual stream writes xp so, every consumer reads it so
bool chain_xp_tiled();
// stream / counters of the next launch; returns < 0 (error set) when the chain has run out of counters
int chain_sync_next(ChainLaunch* out);
// the launch took place; `arrivals` = its workgroups: the gate of the next launch (other stream) is launched with this target.
// Returns < 0 on error.
int chain_sync_done(u32 arrivals);
