"""DVFS experiment (never a reported number): bench.py's MIPS half over an index whose rows are all zero or 7/8 zero, to see what the data's
toggle rate does to the shader clock (MI355X_MICROARCH.md "DVFS give-back").  The switch lives here, not in bench.py: the
driver's benchmark has no knob that changes its data.   usage: python tools/bench_dvfs_data.py zero|sparse [bench.py flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

mode = sys.argv.pop(1)
assert mode in ("zero", "sparse")
_rows = bench.synth_rows


def synth_rows(lo, hi, seed=1234):
    for block in _rows(lo, hi, seed):
        if mode == "zero":
            block.zero_()
        else:                                            # 7/8 of the k-groups zero: low toggle rate, scores stay distinct
            block.view(-1, bench.DIM // 8, 8)[:, 1:, :] = 0
        yield block


bench.synth_rows = synth_rows
if "--no-e2e" not in sys.argv:
    sys.argv.append("--no-e2e")
bench.main()
