// sharded_protocol_test.cpp — the host-side decisions of the sharded collision tick (csrc/sharded_protocol.h, used by
// tick_sharded.hip export_ticks and tick_single.hip wait_for_progress) driven through a MODEL of R ranks without a GPU:
// every rank has a host that issues launches (at most L ahead of its device, deciding on pinned mirror words it reads at
// arbitrary moments, sometimes one decision stale — what mrs_swarm_debug_chaos does to the real hosts) and a device that runs the
// launches in order, each one behind the collective of the previous tick.  Reports (warning W: most of a skin used up; stall T:
// the lists are not good beyond step T) are raised by scripted launches on scripted ranks and reach the other ranks' mirrors
// `vis` launches later (1: serial protocol; up to 5: split protocol with announced stall indices T = launch + 4).
// What must hold whatever the interleaving:
//   * all hosts issue the SAME number of launches in the segment (else the collectives would not match up: a hang on real hardware);
//   * no launch after T runs on any rank — every device knows T before it starts launch T + 1;
//   * all hosts agree on T, W, the ticks that ran and whether to search.
// This models src/multirotor_simulator.cpp:211-217, 295-359 only in so far as the ORDER of ticks and searches goes; physics is not here.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../mrs_multirotor_simulator_amd/csrc/sharded_protocol.h"

using namespace mrs_protocol;

namespace {
bool g_quiet = false;  // negative control: failures are expected, not printed
#define REPORT(...) do { if (!g_quiet) std::printf(__VA_ARGS__); } while (0)
struct Rng {
  unsigned long long s;
  unsigned next() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (unsigned)(s >> 11);
  }
  unsigned below(unsigned n) { return next() % n; }
};

inline unsigned min_nz(unsigned a, unsigned b) { return a == 0u ? b : (b == 0u ? a : (a < b ? a : b)); }

struct Report { int rank; unsigned launch; bool stall; };  // raised DURING that launch (if it runs)

struct Rank {
  // host
  unsigned issued = 0, last = 0;
  bool     done = false;
  unsigned staleT = 0, staleW = 0;
  // device
  unsigned started = 0, finished = 0;  // launches started / finished (no-ops included)
  unsigned hdrT = 0, hdrW = 0;         // own header words (what the next collective carries)
  unsigned mirT = 0, mirW = 0, mirP = 0;  // pinned host mirror
  std::vector<unsigned> sentT, sentW;  // header snapshot at the end of launch k (index k)
  std::vector<bool>     ran;           // launch k really stepped
};

// one segment of n_ticks launches at most; returns false on a violated property
bool run_segment(int R, unsigned lead, bool split, unsigned vis, unsigned horizon, unsigned n_ticks, const std::vector<Report>& reports, unsigned long long seed,
                 bool chaos) {
  Rng                rng{seed * 0x9E3779B97F4A7C15ull + 1};
  const unsigned     ahead = search_ahead(lead, split);
  std::vector<Rank>  rk((size_t)R);
  for (auto& r : rk) {
    r.last = n_ticks;
    r.sentT.assign(n_ticks + 2, 0u);
    r.sentW.assign(n_ticks + 2, 0u);
    r.ran.assign(n_ticks + 2, false);
  }
  // a report raised in launch k becomes part of what every rank folds at the start of launch k + vis (serial: the collective of tick
  // k carries it, the launch after folds it; split: boundary / interior chains, DESIGN §5 "why 4")
  auto known_at_start = [&](int r, unsigned k, unsigned& T, unsigned& W) {
    T = rk[(size_t)r].hdrT;
    W = rk[(size_t)r].hdrW;
    if (k > vis)
      for (int q = 0; q < R; q++) {
        T = min_nz(T, rk[(size_t)q].sentT[k - vis]);
        W = min_nz(W, rk[(size_t)q].sentW[k - vis]);
      }
  };
  unsigned long steps = 0;
  for (;;) {
    bool all_done = true;
    for (auto& r : rk) all_done = all_done && r.done && r.finished == r.issued;
    if (all_done) break;
    if (++steps > 2000000ul) {
      REPORT("FAIL: no progress (deadlock) seed %llu\n", seed);
      return false;
    }
    const int  r = (int)rng.below((unsigned)R);
    Rank&      me = rk[(size_t)r];
    if (rng.below(2) == 0) {  // ---- host r takes a decision ----
      if (me.done) continue;
      const unsigned next = me.issued + 1;
      if (next > me.last) { me.done = true; continue; }
      if (host_is_behind(next, me.mirP, me.mirT, (int)lead)) continue;  // (wait_for_progress spins)
      unsigned T = me.mirT, W = me.mirW;
      if (chaos && rng.below(2) == 0) { const unsigned t = T, w = W; T = me.staleT; W = me.staleW; me.staleT = t; me.staleW = w; }
      else { me.staleT = T; me.staleW = W; }
      me.last = segment_last(me.last, T, W, lead, ahead);
      if (next > me.last) { me.done = true; continue; }
      me.issued = next;
    } else {  // ---- device r advances ----
      if (me.started == me.finished) {  // start the next launch?
        const unsigned k = me.started + 1;
        if (k > me.issued) continue;
        // launch k runs behind collective k - 1 (serial form; in the split form the boundary chain does, and the interior chain is
        // tied to it within one launch): every rank must have finished launch k - 1
        bool ready = true;
        for (int q = 0; q < R; q++) ready = ready && rk[(size_t)q].finished + 1 >= k;
        if (!ready) continue;
        unsigned T, W;
        known_at_start(r, k, T, W);
        me.hdrT = T; me.hdrW = W;
        if (T) me.mirT = T;
        if (W) me.mirW = W;
        me.started = k;
        me.ran[k] = !(T != 0u && k > T);
        if (me.ran[k]) me.mirP = k;
      } else {  // finish the running launch: scripted reports, then the header snapshot the collective carries
        const unsigned k = me.started;
        if (me.ran[k])
          for (const auto& rep : reports)
            if (rep.rank == r && rep.launch == k) {
              if (rep.stall) {
                const unsigned T = k + horizon;  // exact (horizon 0) or announced ahead
                me.hdrT = min_nz(me.hdrT, T);
                if (horizon == 0u) me.mirT = min_nz(me.mirT, T);  // an exact report of a full launch goes to the host at once
              } else {
                me.hdrW = min_nz(me.hdrW, k);
              }
            }
        me.sentT[k] = me.hdrT;
        me.sentW[k] = me.hdrW;
        me.finished = k;
      }
    }
  }
  // ---- properties ----
  for (int r = 1; r < R; r++)
    if (rk[(size_t)r].issued != rk[0].issued) {
      REPORT("FAIL: ranks issued %u and %u launches (seed %llu, split %d)\n", rk[0].issued, rk[(size_t)r].issued, seed, (int)split);
      return false;
    }
  // the final fold of a segment (one more exchange in the split protocol): every rank ends with the same words
  unsigned T = 0, W = 0;
  for (int r = 0; r < R; r++) { T = min_nz(T, rk[(size_t)r].hdrT); W = min_nz(W, rk[(size_t)r].hdrW); }
  const unsigned launched = rk[0].issued;
  for (int r = 0; r < R; r++)
    for (unsigned k = 1; k <= launched; k++) {
      const bool should = !(T != 0u && k > T);
      if (rk[(size_t)r].ran[k] != should) {
        REPORT("FAIL: rank %d launch %u ran=%d but T=%u (seed %llu, split %d, vis %u)\n", r, k, (int)rk[(size_t)r].ran[k], T, seed, (int)split, vis);
        return false;
      }
    }
  const unsigned ran = ticks_ran(T, 1u, launched);
  if (T != 0u && T <= launched && ran != T) { REPORT("FAIL: ticks_ran %u vs T %u\n", ran, T); return false; }
  if (T == 0u && ran != launched) { REPORT("FAIL: ticks_ran %u vs launched %u\n", ran, launched); return false; }
  if (W != 0u && T == 0u && launched != (W + ahead - 1u < n_ticks ? W + ahead - 1u : n_ticks)) {
    REPORT("FAIL: a warning at %u ended the segment at %u, expected %u (seed %llu)\n", W, launched, W + ahead - 1u, seed);
    return false;
  }
  if (T != 0u && launched > T + lead + 1u) { REPORT("FAIL: %u launches issued beyond stall %u + lead\n", launched, T); return false; }
  (void)search_due;
  return true;
}
}  // namespace

int main() {
  // unit checks of the decision functions
  if (host_is_behind(5, 1, 0, 3) != true || host_is_behind(5, 2, 0, 3) != false) { std::printf("FAIL behind\n"); return 1; }
  if (host_is_behind(9, 4, 4, 3) != false || host_is_behind(9, 3, 4, 3) != true) { std::printf("FAIL behind/stall\n"); return 1; }
  if (segment_last(100, 0, 0, 3, 6) != 100 || segment_last(100, 10, 0, 3, 6) != 14 || segment_last(100, 0, 10, 3, 6) != 15 || segment_last(12, 10, 10, 3, 6) != 12) {
    std::printf("FAIL segment_last\n");
    return 1;
  }
  if (ticks_ran(0, 5, 7) != 7 || ticks_ran(7, 5, 7) != 3 || ticks_ran(20, 5, 7) != 7 || ticks_ran(4, 5, 7) != 0) { std::printf("FAIL ticks_ran\n"); return 1; }
  if (!search_due(3, 0, false) || !search_due(0, 3, true) || search_due(0, 3, false) || search_due(0, 0, true)) { std::printf("FAIL search_due\n"); return 1; }
  if (split_residency_ok(0xFFFFFFFFu, 10, 2048, 0) || !split_residency_ok(200, 150, 2048, 0) || split_residency_ok(1024, 150, 2048, 0) ||
      split_residency_ok(200, 513, 2048, 0) || !split_residency_ok(5000, 5000, 2048, 16)) {
    std::printf("FAIL residency\n");
    return 1;
  }
  std::printf("ok functions\n");

  Rng      pick{12345};
  unsigned cases = 0;
  for (int rep = 0; rep < 4000; rep++) {
    const int      R     = 2 + (int)pick.below(7);      // 2..8 ranks
    const unsigned lead  = 1 + pick.below(4);           // MRS_FUSED_LEAD
    const bool     split = pick.below(2) == 1;
    const unsigned vis   = split ? 3 + pick.below(3) : 1;  // launches until a report is in every rank's words
    const unsigned hor   = split ? 4u : 0u;                // announced stall indices in the split protocol (MRS_PRED_HORIZON)
    const unsigned n     = 5 + pick.below(60);
    std::vector<Report> reports;
    const unsigned nrep = pick.below(4);
    for (unsigned k = 0; k < nrep; k++) reports.push_back(Report{(int)pick.below((unsigned)R), 1 + pick.below(n), pick.below(3) == 0});
    if (!run_segment(R, lead, split, vis, hor, n, reports, 1000 + (unsigned long long)rep, rep % 2 == 1)) return 1;
    cases++;
  }
  std::printf("ok model %u segments\n", cases);
  // negative control: the SERIAL constants (search queued L + 3 launches behind a warning, exact stall indices) with reports that
  // take as long as in the split protocol — the model must notice (ranks disagree on the launch count, or a launch runs beyond T)
  g_quiet = true;
  unsigned caught = 0;
  for (int rep = 0; rep < 300; rep++) {
    std::vector<Report> reports{Report{rep % 3, 4u + (unsigned)(rep % 7), rep % 2 == 0}};
    if (!run_segment(4, 3, /*split=*/false, /*vis=*/5, /*horizon=*/0, 40, reports, 7000 + (unsigned long long)rep, true)) caught++;
  }
  g_quiet = false;
  if (caught < 100) { std::printf("FAIL: the negative control was caught only %u times of 300\n", caught); return 1; }
  std::printf("ok negative_control %u of 300\n", caught);
  return 0;
}
