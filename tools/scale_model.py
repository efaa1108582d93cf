"""The per-link model of DESIGN.md section 6, as arithmetic: config 5 (125M x 1.25B rows per GPU) on N GPUs, step = max(wire,
kernels) + head and tail, for a set of link rates; with and without the sender-side precheck of the chunk form.  Inputs are the
one-GPU measurements under profiles/r04_* (kernel milliseconds per rank and step) - nothing here was measured on more than one GPU.
usage: python tools/scale_model.py [hit_fraction=0.5]"""
import sys

hits = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
ROWS_B, ROWS_P = 125e6, 1.25e9
SINGLE_GPU_GPS = 117.9e9               # c3 on one GPU (profiles/r04_c3_bench.json: 8.48-8.8 ms)
WIRE_B_PER_KEY = 7.017                 # tools/pack_probe.py
KERNELS_MS = 13.4                      # per rank and step without the precheck (profiles/r04_c5_one_rank_kernel_stats.csv)
HEAD_TAIL_MS = 2.5                     # first pack before the wire starts + last piece's pass and the join after it ends
FILTER_MS = 4 * 2.3                    # fj_part_filter_inplace at the 8-rank plan, four pieces (profiles/r04_precheck_probe_kernel_stats.txt)
REST_SCALING_MS = 9.6                  # copy into the wire format + owner's pass + probe side of the join: scale with what survives
FALSE_POSITIVES = 0.031                # share of the misses that pass the filters
FILTER_BYTES_PER_BUILD_KEY = 1.07


def step(n, link, precheck):
    peers = n - 1
    f = hits + FALSE_POSITIVES * (1 - hits) if precheck else 1.0
    keys_per_link = (ROWS_P * f + ROWS_B) / n                       # what one rank sends to ONE peer
    wire = keys_per_link * WIRE_B_PER_KEY / link * 1e3
    kern = KERNELS_MS
    head = HEAD_TAIL_MS
    if precheck:
        wire += ROWS_B * FILTER_BYTES_PER_BUILD_KEY / link * 1e3     # every owner's filters to every rank: nb_total / n bytes per link
        kern += FILTER_MS - (1 - f) * REST_SCALING_MS
        head += 1.0
    return (max(wire, kern) if peers else kern) + head, wire, kern


print(f"config 5 per GPU: 125M x 1.25B rows, {hits * 100:.0f} % hits; single GPU {SINGLE_GPU_GPS / 1e9:.1f} G probes/s; step = max(wire, kernels) + head/tail")
print(f"{'N':>2} {'link GB/s':>9} | {'plain: wire':>11} {'kernels':>8} {'step':>7} {'speed-up':>8} | {'precheck: wire':>14} {'kernels':>8} {'step':>7} {'speed-up':>8}")
for n in (2, 4, 8):
    for link in (45e9, 55e9, 65e9):
        a, aw, ak = step(n, link, False)
        b, bw, bk = step(n, link, True)
        sa, sb = n * ROWS_P / (a * 1e-3) / SINGLE_GPU_GPS, n * ROWS_P / (b * 1e-3) / SINGLE_GPU_GPS
        print(f"{n:>2} {link / 1e9:>9.0f} | {aw:>9.1f} ms {ak:>5.1f} ms {a:>5.1f}ms {sa:>7.2f}x | {bw:>12.1f} ms {bk:>5.1f} ms {b:>5.1f}ms {sb:>7.2f}x")
