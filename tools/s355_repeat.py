"""The bench's 355-state measurement REPS times in one process (profiles/r04_hw_queues.txt: with the previous measurement's batches destroyed before the next
are built -- bench._release -> RestartGroups.close -- every run finds the hardware queues free)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = argparse.Namespace(segments=50000, clones=3, update_iters=5, restarts=16)
for rep in range(int(os.environ.get("REPS", 5))):
    rs, S, N1, dt, elbo, prof = bench._timed_run(args, 0, 16, 2, 12, 10, 2)
    top = sorted(prof.items(), key=lambda kv: -kv[1][0])[:5]
    print('run %d: %.1f EM it/s, %.1f ms per step, paced %s | ' % (rep, 16 * 10 / dt, dt / 10 * 1e3, rs.paced) + '  '.join('%s %.2f ms x %d' % (k, v[0] / max(v[1], 1), v[1]) for k, v in top), flush=True)
    bench._release(rs)
