"""Where a round of the H-step goes: kernel durations and the gaps between them.

    workload:  python tools/hstep_gaps.py run            (E-step, then the H-step alone, a few times)
    reduce:    python tools/hstep_gaps.py reduce <kernel_trace.csv>

Under `rocprofv3 --kernel-trace` the reduce step reads start / end of every hstep_lr_tables and hstep_round_* dispatch:
tables duration, the gap to the round kernel, the round's duration, and the turnaround from the end of a round to the
start of the next round's first kernel (mailbox -> SciPy step -> two launches)."""
import csv, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import bench
    from vlgp_amd import engine as E
    from vlgp_amd.api import FitSession
    trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
    sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
    for _ in range(6):
        sess.em_iteration()
    import time
    for _ in range(5):
        E.estep(sess.segs, sess.params, sess.config)
        t0 = time.perf_counter()
        E.hstep(sess.segs, sess.params, sess.config)
        print("H-step alone %.3f ms" % (1e3 * (time.perf_counter() - t0)))
    sess.close()


def reduce(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "hstep_lr_tables" in n or "hstep_round" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "tab" if "tables" in n else "round"))
    rows.sort()
    tab, gap_tr, rnd, turn = [], [], [], []
    for i, (s, e, k) in enumerate(rows):
        if k == "tab":
            tab.append(e - s)
            if i + 1 < len(rows) and rows[i + 1][2] == "round":
                gap_tr.append(rows[i + 1][0] - e)
        else:
            rnd.append(e - s)
            if i + 1 < len(rows):
                d = rows[i + 1][0] - e
                if d < 200000:  # (the next H-step's first round comes milliseconds later)
                    turn.append(d)
    f = lambda x: "%.1f us (median %.1f, n %d)" % (np.mean(x) / 1e3, np.median(x) / 1e3, len(x)) if x else "-"
    print("tables kernel      ", f(tab))
    print("gap tables -> round", f(gap_tr))
    print("round kernel       ", f(rnd))
    print("turnaround to next ", f(turn))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else reduce(sys.argv[2])
