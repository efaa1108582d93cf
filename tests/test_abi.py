"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
include/flashjoin.h declares, the Python mirror exports the reference's names with the reference's
keyword arguments, host-side validation works, and without a GPU the product fails LOUDLY
(no CPU fallback)."""
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, has_gpu


@pytest.fixture(scope="module")
def lib():
    from flash_hash_join_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_native()
    return _lib.load()


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # prose in comments may mention other names
    return sorted(set(re.findall(r"\b(fj_[a-z0-9_]+)\s*\(", hdr)))


def _exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return sorted(l.split()[-1] for l in out.splitlines() if " T " in l)


def test_library_exports_exactly_the_declared_symbols(lib):
    """The drop-in boundary is include/flashjoin.h: 40 entry points, and they are ALL libflashjoin_hip.so exports (header ->
    library and library -> header; the export list csrc/exports.map and the binding's table agree).  The building blocks behind
    it (include/flashjoin_lab.h) are visible in libflashjoin_hip_lab.so only - the same objects linked without the export list."""
    from flash_hash_join_amd import _lib
    public, lab = _declared("flashjoin.h"), _declared("flashjoin_lab.h")
    assert len(public) == 40 and not set(public) & set(lab)
    assert sorted(_lib.SYMBOLS) == public and sorted(_lib.LAB_SYMBOLS) == lab
    m = open(os.path.join(_lib.CSRC, "exports.map")).read()
    assert sorted(re.findall(r"\b(fj_[a-z0-9_]+);", re.sub(r"/\*.*?\*/", "", m, flags=re.S))) == public
    assert _exported(_lib.PRODUCT_LIB_PATH) == public, "libflashjoin_hip.so must export include/flashjoin.h and nothing else"
    assert _exported(_lib.LAB_LIB_PATH) == sorted(public + lab)
    for name in public:
        assert hasattr(lib, name), name
    assert lib.fj_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define FJ_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "flashjoin.h")).read()).group(1))
    # both libraries are links of the SAME objects: every public function has the same size in both
    def sizes(path):
        out = subprocess.check_output(["nm", "-D", "-S", "--defined-only", path], text=True)
        return {l.split()[-1]: l.split()[1] for l in out.splitlines() if " T " in l}
    sp, sl = sizes(_lib.PRODUCT_LIB_PATH), sizes(_lib.LAB_LIB_PATH)
    assert all(sp[n] == sl[n] for n in public)


def test_abi_structs_carry_their_size_and_old_option_variables_are_called_out(lib):
    """structs the library fills or reads on a caller's behalf begin with struct_size (a caller built against another
    layout is refused or served within its own bounds), FJ_OPTIONS rejects non-numeric values instead of reading them as 0, and the
    per-option environment variables that round 5 removed produce one warning instead of being ignored silently."""
    from flash_hash_join_amd import _lib
    assert _lib.FjDistTimings._fields_[0][0] == "struct_size" and _lib.FjDistEngineOps._fields_[0][0] == "struct_size"
    code = "from flash_hash_join_amd import _lib, api; print(api.get_option('join_wide'), api.get_option('radix_threshold'))"
    env = dict(os.environ, FJ_OPTIONS="join_wide=abc,radix_threshold=77,=3,bogus=1", FJ_RADIX_THRESHOLD="5")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["2", "77"]                                  # join_wide=abc ignored (default 2), radix_threshold=77 taken
    assert "ignoring 'join_wide=abc' (want name=integer)" in out.stderr and "ignoring 'bogus=1'" in out.stderr and "ignoring '=3'" in out.stderr
    assert "FJ_RADIX_THRESHOLD is no longer read" in out.stderr


def test_key_mixer_is_a_bijection_and_matches_its_numpy_restatement(lib):
    """fj_key_mix64 / fj_key_unmix64 (host functions of the C ABI: no GPU needed): chunk pools and wire chunks hold the mixed key,
    so the mixer must be one-to-one - unmix(mix(k)) == k, mix(unmix(h)) == h - and tests/keymix.py, which the GPU tests check
    placements and wire chunks against, must be the same function.  Hash word 1 (the high word) is what rounds 1-3 partitioned
    on: its known answers must not move."""
    import keymix
    rng = np.random.default_rng(7)
    ks = np.concatenate([np.array([0, 1, 2, 42, 2**64 - 1, 0x9E3779B97F4A7C15, 1 << 32, (1 << 32) - 1], dtype=np.uint64),
                         rng.integers(0, 2**63, 2000, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 2000, dtype=np.uint64),
                         np.arange(500, dtype=np.uint64) << np.uint64(32), np.arange(500, dtype=np.uint64)])
    m = keymix.mix(ks)
    for k, h in zip(ks.tolist(), m.tolist()):
        assert lib.fj_key_mix64(k) == h and lib.fj_key_unmix64(h) == k
        assert lib.fj_key_mix64(lib.fj_key_unmix64(k)) == k           # onto as well: every 64-bit word is some key's image
    assert np.array_equal(keymix.unmix(m), ks) and np.unique(m).size == np.unique(ks).size

    def w1_round3(k):                                          # fj_hash_w1 as rounds 1-3 defined it
        lo = (k & np.uint64(0xFFFFFFFF)).astype(np.uint32); hi = (k >> np.uint64(32)).astype(np.uint32)
        with np.errstate(over="ignore"):
            x = (lo * np.uint32(0x9E3779B1)) ^ (hi * np.uint32(0x85EBCA77))
            x ^= x >> np.uint32(16); x *= np.uint32(0x85ebca6b)
            x ^= x >> np.uint32(13); x *= np.uint32(0xc2b2ae35)
            x ^= x >> np.uint32(16)
        return x
    assert np.array_equal(keymix.hash_w1(ks), w1_round3(ks))
    # quality: sequential keys and keys that differ in their high word only spread like random ones over 1024 partitions
    for keys in (np.arange(1 << 20, dtype=np.uint64), np.arange(1 << 20, dtype=np.uint64) << np.uint64(32)):
        h = keymix.mix(keys)
        cnt = np.bincount((h >> np.uint64(54)).astype(np.int64), minlength=1024)
        assert cnt.min() > 800 and cnt.max() < 1250               # 1024 +- 4 sigma
        slots = np.bincount((h & np.uint64(8191)).astype(np.int64), minlength=8192)
        assert slots.min() > 70 and slots.max() < 190              # 128 +- 5 sigma


def test_header_is_plain_c_and_the_c_example_links(lib, tmp_path):
    """include/flashjoin.h must be usable from C (the boundary a cgo / JNI / ctypes binding sees): it compiles as C99 and as
    C++11 on its own, and examples/host_join.c builds and links against the library (running it needs a GPU)."""
    from flash_hash_join_amd import _lib
    for h in ("flashjoin.h", "flashjoin_lab.h"):
        hdr = os.path.join(ROOT, "include", h)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
        subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    exe = str(tmp_path / "host_join")
    libdir = os.path.dirname(_lib.PRODUCT_LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "host_join.c"),
                           "-L" + libdir, "-lflashjoin_hip", "-Wl,-rpath," + libdir, "-o", exe])
    assert os.path.exists(exe)
    if not has_gpu():                                           # no GPU: the example must fail loudly through fj_last_error, not crash
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 1 and "flash_join:" in out.stderr


def test_library_contains_gfx950_code_object(lib):
    from flash_hash_join_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"fj_partition_kernel" in blob and b"fj_lds_join_kernel" in blob


def test_python_module_mirrors_reference_exports():
    import flash_join
    from flash_hash_join_amd import api
    assert len(api.REFERENCE_EXPORTS) == 13
    for name in api.REFERENCE_EXPORTS + api.ALIASES:
        assert callable(getattr(flash_join, name)), name
    for name in api.REFERENCE_EXPORTS:
        if name == "initialize":
            continue
        params = list(inspect.signature(getattr(flash_join, name)).parameters)
        assert params[:3] == ["build_keys", "build_values", "probe_keys"], name      # py::arg names, hash_join.cpp:605
    assert flash_join.__doc__.startswith("A high-performance hash join library")     # hash_join.cpp:600


def test_host_argument_normalisation():
    from flash_hash_join_amd.api import _as_u64_host
    a = _as_u64_host(np.array([-1, 2], dtype=np.int64), "x")
    assert a.dtype == np.uint64 and int(a[0]) == 2**64 - 1                           # bit reinterpretation
    b = _as_u64_host(np.arange(10, dtype=np.uint64)[::2], "x")                       # strided: made contiguous, not misread
    assert b.tolist() == [0, 2, 4, 6, 8]
    c = _as_u64_host(np.arange(6, dtype=np.uint64).reshape(2, 3), "x")               # N-D flattened
    assert c.shape == (6,)
    d = _as_u64_host(np.array([1, 2], dtype=np.int32), "x")
    assert d.dtype == np.uint64
    with pytest.raises(TypeError):
        _as_u64_host(np.array(["a"]), "x")


def test_sort_pairs_orders_as_unsigned_on_host_and_torch():
    """The comparison helper for materialised pairs (output order of the joins is unspecified): (key, value) order as
    uint64 - keys above 2^63 (negative as int64) sort last - identical for NumPy arrays and torch tensors."""
    import torch
    import flash_join
    rng = np.random.default_rng(5)
    k = rng.integers(0, 1 << 64, 5000, dtype=np.uint64); k[:2000] = k[2000:4000]           # duplicates: values break the tie
    v = rng.integers(0, 1 << 64, 5000, dtype=np.uint64)
    sk, sv = flash_join.sort_pairs(k, v)
    o = np.lexsort((v, k))
    assert np.array_equal(sk, k[o]) and np.array_equal(sv, v[o]) and sk.dtype == np.uint64
    tk, tv = flash_join.sort_pairs(torch.from_numpy(k.view(np.int64)), torch.from_numpy(v.view(np.int64)))
    assert np.array_equal(tk.numpy().view(np.uint64), sk) and np.array_equal(tv.numpy().view(np.uint64), sv)
    e = flash_join.sort_pairs(np.empty(0, np.uint64), np.empty(0, np.uint64))
    assert e[0].size == 0 and e[1].size == 0


def test_length_mismatch_raises_before_any_device_work(lib):
    import flash_join
    with pytest.raises(ValueError):
        flash_join.hash_join_count(np.arange(4, dtype=np.uint64), np.arange(3, dtype=np.uint64), np.arange(4, dtype=np.uint64))


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_a_gpu(lib):
    import flash_join
    assert lib.fj_device_count() == 0
    with pytest.raises(RuntimeError, match="HIP device"):
        flash_join.initialize()
    k = np.arange(8, dtype=np.uint64)
    with pytest.raises(RuntimeError):
        flash_join.hash_join_count(k, k, k)          # no silent CPU path


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "flash_hash_join_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower().replace("test-suite with a stand-in engine", ""), os.path.join(dirpath, f)
    assert "oracle" not in open(os.path.join(ROOT, "flash_join.py")).read().lower()


def test_missing_library_is_an_import_error(tmp_path, monkeypatch):
    from flash_hash_join_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_dispatch_options_round_trip(lib):
    """fj_set_option / fj_get_option are host-only: defaults, round trip, unknown names and bad values are errors."""
    from flash_hash_join_amd import api
    assert api.get_option("radix_threshold") == 0 and api.get_option("scalar_hbm_table") == 0
    assert api.get_option("persistent_min_items") == 8192
    assert api.get_option("bloom_auto") == 1 and api.get_option("bloom_auto_max_hit_bp") == 2300 and api.get_option("plan_target_keys") == 4096
    try:
        api.set_option("radix_threshold", 123456); assert api.get_option("radix_threshold") == 123456
        api.set_option("scalar_hbm_table", 7); assert api.get_option("scalar_hbm_table") == 1
        api.set_option("persistent_min_items", 0); assert api.get_option("persistent_min_items") == 0
        with pytest.raises(RuntimeError, match="unknown option"):
            api.set_option("overlap_relations", 1)          # the two-stream schedule of rounds 1-2 is gone
        with pytest.raises(RuntimeError, match="bloom_variant"):
            api.set_option("bloom_variant", 3)
        assert api.get_option("bloom_variant") in (0, 1, 2)
        api.set_option("bloom_auto", 0); assert api.get_option("bloom_auto") == 0
        api.set_option("plan_target_keys", 64); assert api.get_option("plan_target_keys") == 64
        with pytest.raises(RuntimeError, match="plan_target_keys"):
            api.set_option("plan_target_keys", 5000)
        with pytest.raises(RuntimeError, match="unknown option"):
            api.set_option("no_such_option", 1)
        with pytest.raises(RuntimeError, match=">= 0"):
            api.set_option("radix_threshold", -1)
        with pytest.raises(KeyError):
            api.get_option("no_such_option")
    finally:
        api.set_option("radix_threshold", 0); api.set_option("scalar_hbm_table", 0); api.set_option("persistent_min_items", 8192)
        api.set_option("bloom_auto", 1); api.set_option("plan_target_keys", 4096)


def test_multi_gpu_strategy_model():
    """What a multi-rank join runs as is pure host arithmetic: the environment names a strategy, or (unset: "auto") the C++ driver's
    cost model weighs the owner shuffle against the build broadcast for a counting step's sizes (fj_dist_model: the same verdict on
    every rank) - probe-heavy joins broadcast the build side, balanced ones shuffle; the plan mirror matches the native planner's
    documented break points."""
    from flash_hash_join_amd import distributed as D
    os.environ.pop("FJ_DIST_STRATEGY", None)
    assert D.choose_strategy() == "auto"
    for forced in ("shuffle", "broadcast", "scatter", "auto"):
        os.environ["FJ_DIST_STRATEGY"] = forced
        assert D.choose_strategy() == forced
    for gone in ("bogus", "replicate"):                        # (rounds 1-4's unpartitioned replicate is superseded by the broadcast form and gone)
        os.environ["FJ_DIST_STRATEGY"] = gone
        with pytest.raises(ValueError):
            D.choose_strategy()
    os.environ.pop("FJ_DIST_STRATEGY")
    # BASELINE configs[4]'s per-rank sizes: every link carries 0.75 GB (broadcast) against 9.7 / 2.4 / 1.2 GB (shuffle at N = 2 / 4 / 8)
    for world in (2, 4, 8):
        for rate in (45e9, 55e9, 65e9):
            m = D.form_model(world, 125_000_000, 1_250_000_000, rate)
            assert m["pick"] == "broadcast" and m["broadcast"] < 0.021 and m["shuffle"] > 0.021, (world, rate, m)
    m = D.form_model(8, 125_000_000, 1_250_000_000, 55e9)
    assert 0.0155 < m["broadcast"] < 0.0159 and 0.023 < m["shuffle"] < 0.026, m           # wire-bound at ~15.7 ms in the 8 pieces the driver then takes (16.3 in 4; kernels: 14.8) against a wire-bound ~24.4 ms
    assert 0.0145 < D.form_model(8, 125_000_000, 1_250_000_000, 70e9)["broadcast"] < 0.0153      # kernel-bound from ~62 GB/s up: pack 1.4 + passes 8.6 + join 4.95 ms
    # materialising joins: the regions carry the values (14 bytes per build row) and the join is the plain kernel - the broadcast
    # still wins where one or three links would carry the shuffle, the shuffle at 8 ranks
    assert [D.form_model(w, 125_000_000, 1_250_000_000, 55e9, materialize=True)["pick"] for w in (2, 4, 8)] == ["broadcast", "broadcast", "shuffle"]
    assert D.form_model(8, 500_000_000, 500_000_000, 55e9)["pick"] == "shuffle"          # as many build rows as probe rows: the regions outweigh the rows
    m = D.form_model(8, 125_000_000, 1_250_000_000, 4000e9)                                # links as fast as HBM: kernels only, and the two forms'
    assert abs(m["shuffle"] - m["broadcast"]) < 0.1 * m["shuffle"], m                      # kernels are within a tenth of each other (15.9 / 15.0 ms)
    # the Python layer reads six environment variables and no more
    src = open(os.path.join(ROOT, "flash_hash_join_amd", "distributed.py")).read() + open(os.path.join(ROOT, "flash_hash_join_amd", "_lib.py")).read() + \
        open(os.path.join(ROOT, "flash_hash_join_amd", "api.py")).read() + open(os.path.join(ROOT, "flash_hash_join_amd", "lab.py")).read()
    assert sorted(set(re.findall(r"environ[^\n]*?[\"'](FJ_[A-Z_]+)[\"']", src))) == ["FJ_DIST_NATIVE", "FJ_DIST_NO_FALLBACK", "FJ_DIST_PIECES", "FJ_DIST_PREFILTER", "FJ_DIST_STRATEGY", "FJ_LIB_VARIANT"]
    assert len(open(os.path.join(ROOT, "flash_hash_join_amd", "distributed.py")).read().splitlines()) <= 900


def test_shuffle_prefilter_mode(monkeypatch):
    """The sender-side precheck of the chunk-form shuffle: environment / `bloom` / an explicit override -> mode ("auto" for every
    join across more than one rank: the per-link model of _chunk_prefilter_break_even decides from a sample)."""
    from flash_hash_join_amd import distributed as D
    monkeypatch.delenv("FJ_DIST_PREFILTER", raising=False)
    assert D._chunk_prefilter_mode(False, 1) == "off" and D._chunk_prefilter_mode(True, 1) == "auto" and D._chunk_prefilter_mode(False, 2) == "auto"
    for env, mode in (("0", "off"), ("1", "on"), ("auto", "auto")):
        monkeypatch.setenv("FJ_DIST_PREFILTER", env)
        assert D._chunk_prefilter_mode(False, 1) == D._chunk_prefilter_mode(True, 8) == mode
        assert D._chunk_prefilter_mode(False, 8, "on") == "on"                  # an explicit argument wins
    monkeypatch.setenv("FJ_DIST_PREFILTER", "bogus")
    assert D._chunk_prefilter_mode(True, 1) == "auto" and D._chunk_prefilter_mode(False, 1) == "off"


def test_bench_launches_its_own_ranks_when_asked_for_several_gpus(monkeypatch):
    """`python bench.py --gpus N` (the driver's command line) outside a torchrun environment must start
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before touching the GPU and
    relay its exit code; the default multi-GPU workload is c5 (BASELINE configs[4]: 125M x 1.25B rows per GPU)."""
    import subprocess
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, cwd=None):
            seen["cmd"], seen["env"] = cmd, env

        def wait(self):
            return 7

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert "127.0.0.1" in cmd and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.WORKLOADS["c5"][:2] == (125_000_000, 1_250_000_000)
