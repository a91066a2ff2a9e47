"""Work model of two tilings of the packed attention stacks (CPU, no GPU needed): how many (query, key) score slots the kernels EXECUTE per useful
(query, key) pair of the step's three encoder stacks, for

  per-sequence tiles (what runs: csrc/attention.hip): a wave owns 32 consecutive queries of ONE sequence, walks that sequence's keys in 32-key steps;
      a sequence of n tokens costs ceil(n / 32) waves x ceil(n / 32) steps x 1,024 slots, and workgroups are 4 waves (a ragged last workgroup has idle
      waves that only help to stage);
  stream windows (VERDICT r04 item 4's proposal): the packed token stream is cut into windows of W = 128 / 256 rows regardless of sequence boundaries,
      a wave owns 32 consecutive ROWS (possibly of two sequences), walks the keys of every sequence its rows touch and masks block-diagonally;
      a wave whose rows span sequences a..b costs ceil(keys(a..b) / 32) steps.

usage: python tools/attn_tiling_model.py   (length distributions of bench_e2e.py's synthetic evidence: context tower U[105,171], one-context pass
U[125,200], reader encoder U[300,512]; 3,200 sequences each)"""
import numpy as np

rng = np.random.default_rng(0)
STACKS = (("context tower", 105, 171), ("one-context pass", 125, 200), ("reader encoder", 300, 512))


def per_sequence(lens):
    waves = np.ceil(lens / 32)
    steps = np.ceil(lens / 32)
    slots = (waves * steps * 1024).sum()
    wgs = np.ceil(lens / 128).sum()
    wave_slots = wgs * 4                                   # resident wave slots occupied (idle waves of a ragged workgroup included)
    return slots, waves.sum(), wave_slots


def stream(lens, W):
    cu = np.concatenate([[0], np.cumsum(lens)])
    total = int(cu[-1])
    n_waves = (total + 31) // 32
    starts = np.arange(n_waves) * 32
    ends = np.minimum(starts + 32, total) - 1
    a = np.searchsorted(cu, starts, side="right") - 1      # first / last sequence a wave's rows belong to
    b = np.searchsorted(cu, ends, side="right") - 1
    keys = cu[b + 1] - cu[a]                               # keys of every sequence the wave touches (block-diagonal mask inside)
    slots = (np.ceil(keys / 32) * 1024).sum()
    wgs = (total + W - 1) // W
    return slots, n_waves, wgs * (W // 32)


print("%-18s %10s | %-34s | %-34s | %-34s" % ("stack", "useful", "per-sequence tiles (runs today)", "stream windows of 128 rows", "stream windows of 256 rows"))
for name, lo, hi in STACKS:
    lens = rng.integers(lo, hi + 1, size=3200).astype(np.int64)
    useful = float((lens * lens).sum())
    row = "%-18s %9.2fG |" % (name, useful / 1e9)
    for slots, waves, wslots in (per_sequence(lens), stream(lens, 128), stream(lens, 256)):
        row += " x%.2f slots/pair, %6d waves of %6d |" % (slots / useful, waves, wslots)
    print(row)
print("slots/pair = executed score slots per useful (query, key) pair: 1.00 is perfect; 'waves of N' = working waves / wave slots the workgroups occupy.")
