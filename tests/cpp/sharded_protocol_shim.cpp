// sharded_protocol_shim.cpp — the host-side decision functions of the sharded collision tick (csrc/sharded_protocol.h: what
// tick_sharded.hip export_ticks and tick_single.hip wait_for_progress call) behind a C ABI, so that tests/test_sharded_gloo.py can
// drive them from real PROCESSES bound by gloo collectives.  Built by the test with g++; no GPU, no HIP.
#include "../../mrs_multirotor_simulator_amd/csrc/sharded_protocol.h"

extern "C" {
int      sp_host_is_behind(unsigned index, unsigned P, unsigned T, int lead) { return mrs_protocol::host_is_behind(index, P, T, lead) ? 1 : 0; }
unsigned sp_search_ahead(unsigned lead, int split) { return mrs_protocol::search_ahead(lead, split != 0); }
unsigned sp_segment_last(unsigned last, unsigned T, unsigned W, unsigned lead, unsigned ahead) { return mrs_protocol::segment_last(last, T, W, lead, ahead); }
unsigned sp_ticks_ran(unsigned T, unsigned first, unsigned launched) { return mrs_protocol::ticks_ran(T, first, launched); }
int      sp_search_due(unsigned T, unsigned W, int ticks_left) { return mrs_protocol::search_due(T, W, ticks_left != 0) ? 1 : 0; }
}
