"""Lease-side evidence beside tests/test_gpu_mac_regimes.py: tests/mac_regimes.run_random_batch (1..26 partitions, clips of 1..60
blocks, ragged, static + moving + zero-emitter events, every row against the oracle) over many more seeds than the test suite runs,
at every block size the library supports, on W worker processes that share the GPU.  Prints which accumulate instantiations ran.
    python3 profiles/tools/fuzz_regimes.py FIRST LAST [WORKERS]"""
import collections, os, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
BLOCKS = tuple(int(x) for x in os.environ.get("FUZZ_BLOCKS", "10 11 12 13 14 13").split())     # default: B = 8192, the production block, twice


def work(span):
    from audiblelight_amd import engine
    from tests import mac_regimes as mr
    r = engine.Renderer()
    out = []
    for seed in range(*span):
        lb = BLOCKS[seed % len(BLOCKS)]
        try:
            codes, p, k = mr.run_random_batch(r, seed, log2_block=lb)
            out.append((seed, lb, p, k, tuple(sorted(codes)), None))
        except Exception as exc:  # noqa: BLE001 -- a fuzz driver reports everything
            out.append((seed, lb, -1, -1, (), f"{type(exc).__name__}: {str(exc)[:200]}"))
    return out


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    step = max(1, (last - first + workers * 4 - 1) // (workers * 4))
    spans = [(a, min(last, a + step)) for a in range(first, last, step)]
    t0 = time.time()
    with ProcessPoolExecutor(workers) as pool:
        rows = [row for part in pool.map(work, spans) for row in part]
    bad = [row for row in rows if row[5]]
    hist = collections.Counter(code for row in rows for code in row[4])
    print(f"seeds {first}..{last - 1}: {len(rows)} batches, {len(bad)} failed, {time.time() - t0:.0f} s on {workers} processes")
    print("partitions seen:", sorted({row[2] for row in rows if row[2] > 0}))
    print("largest clip (blocks):", max(row[3] for row in rows))
    for lb in sorted(set(BLOCKS)):
        print(f"log2_block {lb}: {sum(1 for row in rows if row[1] == lb)} batches")
    print("accumulate instantiation -> batches that ran it:")
    for code, n in sorted(hist.items()):
        print(f"  {code:>9} {n}")
    for row in bad:
        print("FAILED", row)
    sys.exit(1 if bad else 0)
