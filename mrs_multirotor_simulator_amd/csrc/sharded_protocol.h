// sharded_protocol.h — the host-side DECISIONS of the sharded collision tick (tick_sharded.hip export_ticks, tick_single.hip
// wait_for_progress) as pure functions, so that they can be exercised without a GPU (tests/cpp/sharded_protocol_test.cpp drives a
// multi-rank model of launches, collectives and pinned mirror words through them under random host skew).
//
// What the protocol rests on: all ranks must issue the same launches and collectives in the same order, yet no host ever
// waits for another host.  Every decision is a function of words that reach all ranks with the positions themselves (the headers of
// the export collective, folded by the launches into pinned host words), taken at a launch index that is a function of those words
// alone:
//   * warning word W (tick in which some UAV of some rank had used most of its skin): the segment ends with launch W + D - 1, the
//     search follows;
//   * stall word T (the lists are not good for evaluating a tick on the state after step T; launches > T are no-ops everywhere):
//     the segment ends with launch T + L + 1;
//   * L = launches a host may run ahead of its device (progress word), D = L + 3 (serial protocol: a report is in every rank's
//     mirror when the launch after it starts) or L + 6 (split protocol: at most five launches later): a host deciding on launch
//     W + D, or on T + L + 2, has provably seen W, or T.
#pragma once

namespace mrs_protocol {

// launches a host may still be behind: `index` is the launch it wants to issue, P the last launch its device has reported as started,
// T the stall index it knows of (0: none).  Launches after T are no-ops and report no progress: once launch T has started nothing
// more will come — but a KNOWN T alone does not end the waiting (a sharded swarm announces stall indices ahead of time; an earlier one
// may still turn up, and the host must not outrun what it has seen).
inline bool host_is_behind(unsigned index, unsigned P, unsigned T, int lead) { return (int)(index - P) > lead && !(T != 0u && P >= T); }

// how far ahead of a warning the search is queued (see above)
inline unsigned search_ahead(unsigned lead, bool protocol_split) { return lead + (protocol_split ? 6u : 3u); }

// last launch index of the running segment given what this host knows now (monotone: it only ever shrinks)
inline unsigned segment_last(unsigned last, unsigned T, unsigned W, unsigned lead, unsigned ahead) {
  if (T != 0u && T + lead + 1u < last) last = T + lead + 1u;
  if (W != 0u && W + ahead - 1u < last) last = W + ahead - 1u;
  return last;
}

// ticks of a segment that really ran: `first` = index of its first launch, `launched` = launches issued, T = the stall index every
// rank agrees on after the segment's final fold (launches after T were no-ops on every rank)
inline unsigned ticks_ran(unsigned T, unsigned first, unsigned launched) {
  return (T != 0u && T + 1u >= first && T + 1u - first < launched) ? T + 1u - first : launched;
}

// after a segment: do all ranks search now?  (T: the lists are stale; W: they are about to be — unless the call is over anyway)
inline bool search_due(unsigned T, unsigned W, bool ticks_left) { return T != 0u || (W != 0u && ticks_left); }

// Residency of the split form: the waves that SPIN inside a split tick hold their wave slots while they wait — block 0
// and the layer-1 blocks of an interior launch (for the boundary launch of the previous tick), the boundary blocks (for an interior
// launch).  "Producers are enqueued before consumers" covers the hardware queues, not SIMD and register slots: a rank stays in the
// serial form unless the spinners leave at least half of the wave slots (at the interior kernel's two waves per SIMD) to everybody
// else, or the boundary chain owns compute units of its own.  n_layer1 == 0xFFFFFFFF: the search has not reported yet.
inline bool split_residency_ok(unsigned n_layer1, unsigned n_boundary, int resident_waves, int cu_reserve) {
  if (n_layer1 == 0xFFFFFFFFu) return false;
  if (cu_reserve > 0) return true;
  return (long long)n_layer1 + 1 <= resident_waves / 2 && (long long)n_boundary <= resident_waves / 4;
}

}  // namespace mrs_protocol
