"""The per-link model of DESIGN.md section 6, printed from the arithmetic the C++ driver itself decides with (fj_dist_model, csrc/fj_dist.hip):
config 5 (125M x 1.25B rows per GPU) on N = 2, 4, 8 GPUs in both forms of the multi-GPU step - the owner shuffle and the build broadcast - for a
set of link rates: step = max(bytes per link / link rate, kernel seconds per rank) + what cannot overlap.  The kernel seconds are one-GPU
measurements (profiles/r06_bcast_one_rank_2_4_8.txt: pack, probe-side passes, dense join at the 2- / 4- / 8-rank plans; profiles/r04_c5_one_rank_kernel_stats.csv:
the shuffle's step); nothing here was measured on more than one GPU.  usage: python tools/scale_model.py [nb_rank=125000000] [np_rank=1250000000]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_hash_join_amd import distributed as D

nb, np_ = (int(x) for x in (sys.argv[1:3] + ["125000000", "1250000000"][len(sys.argv) - 1:])[:2])
SINGLE_GPU_GPS = 115.1e9               # c3 on one GPU (BENCH_r05.json, the driver's own run: 8.688 ms per 1B probes)
print(f"{nb} x {np_} rows per rank; one GPU alone: {SINGLE_GPU_GPS / 1e9:.1f} G probes/s (c3)")
print("%-3s %-10s | %-34s | %-34s | pick" % ("N", "link GB/s", "owner shuffle: ms, G probes/s, x", "build broadcast: ms, G probes/s, x"))
for n in (2, 4, 8):
    for rate in (45e9, 55e9, 65e9, 76.8e9):
        m = D.form_model(n, nb, np_, rate)
        cell = lambda t: "%6.2f ms %7.1f G/s %5.2fx" % (t * 1e3, n * np_ / t / 1e9, n * np_ / t / SINGLE_GPU_GPS)
        print("%-3d %-10.1f | %-34s | %-34s | %s" % (n, rate / 1e9, cell(m["shuffle"]), cell(m["broadcast"]), m["pick"]))
print("bytes per link and step: shuffle 7.02 * (np + nb) / N per rank = %.2f / %.2f / %.2f GB at N = 2 / 4 / 8; broadcast 6.01 * nb = %.2f GB at every N"
      % tuple([7.02 * (np_ + nb) / n / 1e9 for n in (2, 4, 8)] + [6.01 * nb / 1e9]))
