"""Worker of tests/test_peer_window_gpu.py (a process of its own: GPU_MAX_HW_QUEUES must be in the environment before the HIP runtime
starts).  `world` virtual shards in THIS process on the one GPU, one host thread each, bound by the peer-window exchange: the ranks'
exchange kernels write into each other's windows and wait for each other ON THE DEVICE — no host copies, no barrier, no collective
library.  Hosts skewed by mrs_swarm_debug_chaos, ticks in the split form, the whole swarm on the oracle (LITERAL, 1e-11).
usage: peer_window_worker.py world n_total chaos_us"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world, n_total, chaos_us = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    assert int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) >= 2 * world
    os.environ["MRS_SHARD_SPLIT_MIN_BLOCKS"] = "1"
    os.environ["MRS_SHARD_SPLIT_MAX_FRACTION"] = "0.95"
    import helpers
    import mrs_multirotor_simulator_amd as M
    from helpers import RTOL_LITERAL
    from oracle import oracle_swarm as oracle
    from test_export_sets_gpu import DT, VirtualShards, moving_swarm
    M.load_library()
    rng = np.random.default_rng(9100 + world)
    pos, st, cmd = moving_swarm(rng, n_total, speed=6.0)
    hot = rng.choice(n_total, 40, replace=False)
    st["v"][hot] = rng.normal(0, 1, (40, 3)) * [12.0, 12.0, 4.0]
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL,
                       M.EXCHANGE_EXPORT_SETS, transport="peer")
    if chaos_us > 0:
        for r, (g, _) in enumerate(vs.shards):
            g.debug_chaos(chaos_us, seed=23 * world + r)
    done = 0
    for n, crash in ((41, False), (1, True), (118, False), (140, False)):  # 300 ticks, one of them in crash mode
        vs.tick_n(n, True, crash, 100.0)
        for _ in range(n):
            o.step_n(DT, 1, 8)
            o.handle_collisions(True, crash, 100.0)
        done += n
        a, so, fo = vs.gather(), o.get_state(), o.get_external_force()
        assert np.array_equal(a["crashed"], o.has_crashed()), f"crash flags after {done} ticks"
        helpers.assert_close(a["f"], fo, 1e-11, f"forces after {done} ticks")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[k], so[k], RTOL_LITERAL, f"{k} after {done} ticks")
        helpers.assert_close_per_uav(a, so, RTOL_LITERAL, f"after {done} ticks")
    assert (np.abs(fo).sum(axis=1) > 0).sum() > 30 and o.has_crashed().sum() > 0
    info, split = vs.info(), [g.split_stats() for g, _ in vs.shards]
    vs.close()
    for ci in info:
        assert ci["ticks"] == 300 and 3 <= ci["searches"] <= 150 and ci["rccl_ranks"] == 0, ci
    assert sum(t for t, _ in split) > 20 * world, split
    print(f"PEER-WINDOW OK world {world}: searches {[ci['searches'] for ci in info]}, replayed {[ci['noop_ticks'] for ci in info]}, split ticks / boundary blocks {split}")


if __name__ == "__main__":
    main()
