"""CPU: the arithmetic behind bench.py's roofline block, without a GPU -- the algorithmic byte count of SURVEY 8d, the merge of
PMC passes into per-keyswitch figures (tools/pmc_summary.derive, which bench.py's in-run passes go through), and the power
sampler's behaviour on a box without the hwmon files."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tools")]


def test_algorithmic_bytes_match_the_survey():
    import bench
    assert bench.ks_alg_bytes(16384, 7) == 4_587_520                  # 917,504 + 2 x 1,835,008
    assert bench.ks_alg_bytes(16384, 6) == 3_932_160


def test_pmc_passes_merge_into_per_keyswitch_figures():
    import pmc_summary
    batch, L = 256, 7
    vals = {
        "k_ksx_main<14, 4, 3, false>": {"SQ_INSTS_VALU": 3.6e8, "FETCH_SIZE": 1.0e6, "WRITE_SIZE": 5.0e5, "GRBM_GUI_ACTIVE": 8 * 2.0e6,
                                         "SQ_WAVE_CYCLES": 2.0e9, "SQ_ACTIVE_INST_ANY": 4.0e8, "SQ_WAIT_INST_ANY": 7.0e8, "SQ_WAIT_ANY": 9.0e8},
        "k_ksx_intt<14, 4, 3, false>": {"SQ_INSTS_VALU": 4.0e7, "FETCH_SIZE": 2.0e5, "WRITE_SIZE": 2.0e5, "GRBM_GUI_ACTIVE": 8 * 2.0e5},
        "k_ntt_fwd_p<14, 4, 3>": {"SQ_INSTS_VALU": 1.0e7, "FETCH_SIZE": 1.0e5, "WRITE_SIZE": 1.0e5},     # not a keyswitch kernel
    }
    dur = {"k_ksx_main<14, 4, 3, false>": [1000.0, 1000.0], "k_ksx_intt<14, 4, 3, false>": [100.0], "k_ntt_fwd_p<14, 4, 3>": [80.0]}
    d = pmc_summary.derive(vals, dur, batch, L, simds=1024)
    assert d["valu_wave_instructions_per_keyswitch"] == (3.6e8 + 4.0e7) / batch
    # FETCH_SIZE in KiB, doubled (gfx950 under-reports these kernels' reads by 2x); WRITE_SIZE in KiB as is
    want = ((1.0e6 * 2 + 5.0e5) + (2.0e5 * 2 + 2.0e5)) * 1024 / batch
    assert abs(d["traffic_bytes_per_keyswitch"] - want) < 1e-6
    assert abs(d["shader_clock_ghz"] - 2.0) < 1e-9                      # the longest kernel's: 2.0e6 cycles per XCD over 1000 us
    k = d["kernels"]["k_ksx_main<14, 4, 3, false>"]
    assert abs(k["fp64_issue_frac"] - 3.6e8 * 4 / 1024 / 2.0e3 / 1000.0) < 1e-9
    assert abs(sum(k["wave_time_split"].values()) - 1.0) < 1e-9
    assert d["alg_bytes_per_keyswitch"] == 4_587_520 and "k_ntt_fwd_p<14, 4, 3>" not in d["kernels"]


def test_power_sampler_without_hwmon_is_silent():
    import bench
    s = bench.PowerSampler(0)
    s.dir = None                                                    # a box without the amdgpu hwmon files
    s.start()
    assert s.stop() is None


def test_alu_fraction_uses_the_timed_regions_clock():
    """round 3's 0.698 divided un-profiled time by the PMC passes' clock; the timed region's own clock gives 0.66"""
    import bench
    vw, cus, us = 1_811_296, 256, 4.966                                  # BENCH_r03: instructions / keyswitch, measured us / keyswitch
    a = bench.alu_block(vw, cus, us, timed_sclk_mhz=2153.0, pmc_clock_ghz=2.042)
    assert abs(a["issue_us_per_keyswitch"] - vw * 4 / 1024 / 2153.0) < 1e-12
    assert abs(a["achieved_frac"] - 0.6617) < 5e-4 and abs(a["achieved_frac_at_pmc_pass_clock"] - 0.6977) < 5e-4
    assert a["achieved_frac"] < a["achieved_frac_at_pmc_pass_clock"] and a["shader_clock_ghz"] == 2.153
    b = bench.alu_block(vw, cus, us, timed_sclk_mhz=None, pmc_clock_ghz=2.042)   # no hwmon: falls back, and says so
    assert b["achieved_frac"] == b["achieved_frac_at_pmc_pass_clock"] and "PMC" in b["shader_clock_source"]


def test_short_kernels_do_not_get_their_own_clock():
    """GRBM_GUI_ACTIVE of a short dispatch also covers the set-up around it (round 4: 3.02 "GHz" for the 165 us k_ksx_special made its
    issue fraction read 0.44): per-kernel clocks outside (0.5, 2.45) GHz fall back to the longest kernel's"""
    import pmc_summary
    vals = {"k_ksx_main<14, 4, 3, false, true>": {"SQ_INSTS_VALU": 3.4e8, "GRBM_GUI_ACTIVE": 8 * 2.1e6},
            "k_ksx_special<14, 4, 3, true>": {"SQ_INSTS_VALU": 5.58e7, "GRBM_GUI_ACTIVE": 4.0e6}}
    dur = {"k_ksx_main<14, 4, 3, false, true>": [1000.0], "k_ksx_special<14, 4, 3, true>": [165.5]}
    d = pmc_summary.derive(vals, dur, 256, 7, simds=1024)
    sp = d["kernels"]["k_ksx_special<14, 4, 3, true>"]
    assert abs(d["shader_clock_ghz"] - 2.1) < 1e-9 and abs(sp["shader_clock_ghz"] - 2.1) < 1e-9 and sp["shader_clock_ghz_raw"] > 3.0
    assert abs(sp["fp64_issue_frac"] - 5.58e7 * 4 / 1024 / 2.1e3 / 165.5) < 1e-9 and 0.6 < sp["fp64_issue_frac"] < 0.65


def test_gpus_flag_is_never_silently_ignored(monkeypatch):
    """`--gpus N` means N ranks: without a launcher bench.py becomes one (torch.distributed.run, 127.0.0.1, one process per GPU --
    the reference's DevicePool of NUM_DEV runners, host/src/fpga.cpp:1646-1673); under a launcher with another WORLD_SIZE it refuses."""
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        class R: returncode = 7
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    try:
        bench.main()
        raise AssertionError("main() must exit with the launcher's status")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [str(ROOT / "bench.py"), "--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher whose world differs from --gpus: refuse before touching a GPU
    for world, gpus in (("1", "2"), ("2", "1"), ("8", "4")):
        monkeypatch.setenv("WORLD_SIZE", world)
        monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", gpus])
        try:
            bench.main()
            raise AssertionError("must refuse")
        except SystemExit as e:
            assert "WORLD_SIZE" in str(e.code) and "refusing" in str(e.code)


def test_key_stream_split_is_a_same_kernel_difference():
    """roofline.key_stream_bytes_per_keyswitch = L2-miss-side bytes of the real pipeline minus those with every key row aliased onto row 0,
    both on the SHIPPED kernel objects (round 5; rounds 3-4 subtracted a pass of the profiling build, whose kernels move 19.2 MB per
    keyswitch where the shipped ones move 14.4, and printed 0.7 MB where ~5 MB is right). The arithmetic, on profiles/r05_fetch_reconcile's
    numbers; and the key-alias library must be made of the shipped objects plus ONE other translation unit."""
    import bench
    d = bench.key_stream_split(14.43e6, 9.41e6)
    assert abs(d["key_stream_bytes_per_keyswitch"] - 5.02e6) < 1 and d["dram_side_estimate_bytes_per_keyswitch"] == 9.41e6
    mk = (ROOT / "hexl-fpga_amd" / "csrc" / "Makefile").read_text()
    tmk = (ROOT / "tools" / "Makefile").read_text()
    assert "OBJS     = $(KOBJS) $(OUT)/alias_knob.o $(OUT)/host_simd.o" in mk
    # round 6: the variants are built by tools/Makefile into tools/lib_var/, never beside the shipped library
    assert "keyalias" not in mk.split("all:")[1].split("\n")[0] and "_prof.so" not in mk.split("all:")[1].split("\n")[0]
    assert "$(VAR)/libhexl_mi355x_keyalias.so: $(KOBJS) $(LIB)/host_simd.o $(VAR)/alias_knob_keys.o" in tmk
    kobjs = [l for l in tmk.splitlines() if l.startswith("KOBJS")][0]
    assert all(f"$(LIB)/{o}.o" in kobjs for o in ("ntt", "dyadic", "keyswitch", "keyswitch_f64", "keyswitch_lat", "keyswitch_x", "capi"))
    knob = (ROOT / "hexl-fpga_amd" / "csrc" / "alias_knob.hip").read_text()
    assert "#ifdef HEXL_KEY_ALIAS_KNOB" in knob and "return 0u;" in knob          # the shipped variant reads no environment variable
    assert "getenv" not in knob.split("#else")[1]


def test_ntt_roofline_block_arithmetic():
    """BASELINE's second metric gets its own roofline block (round 5): HBM fraction from the timed launch, traffic and the FP64-issue
    fraction from the in-run PMC passes (transform kernel + k_ntt_prepare)."""
    import bench
    import pmc_summary
    vals = {"k_ntt_fwd_p<14, 4, 3, false>": {"SQ_INSTS_VALU": 2.3e7, "FETCH_SIZE": 70000.0, "WRITE_SIZE": 131072.0,
                                             "SQ_WAVE_CYCLES": 1.0e9, "SQ_ACTIVE_INST_ANY": 2.2e8, "SQ_WAIT_INST_ANY": 4.1e8, "SQ_WAIT_ANY": 3.7e8},
            "k_ntt_inv_p<14, 4, 3, false>": {"SQ_INSTS_VALU": 2.5e7, "FETCH_SIZE": 70000.0, "WRITE_SIZE": 131072.0},
            "k_ntt_prepare": {"SQ_INSTS_VALU": 1.0e5, "FETCH_SIZE": 128.0, "WRITE_SIZE": 128.0},
            "k_ksx_main<14, 4, 3, false, true, false>": {"SQ_INSTS_VALU": 3.6e8}}
    dur = {"k_ntt_fwd_p<14, 4, 3, false>": [72.0, 74.0], "k_ntt_inv_p<14, 4, 3, false>": [70.0], "k_ntt_prepare": [3.0], "k_ksx_main<14, 4, 3, false, true, false>": [1000.0]}
    d = pmc_summary.derive_ntt(vals, dur, 1024, simds=1024, clock_ghz=2.0)
    f = d["fwd"]
    assert f["alg_bytes_per_launch"] == 1024 * 262144                                  # 262,144 B per transform
    assert f["read_bytes_per_launch"] == (70000.0 + 128.0) * 1024 * 2 and f["write_bytes_per_launch"] == (131072.0 + 128.0) * 1024
    assert abs(f["traffic_over_algorithmic"] - (f["read_bytes_per_launch"] + f["write_bytes_per_launch"]) / (1024 * 262144)) < 1e-12
    assert f["valu_wave_instructions_per_transform"] == 2.3e7 / 1024
    assert abs(f["fp64_issue_frac_under_pmc"] - 2.3e7 * 4 / 1024 / 2000.0 / 73.0) < 1e-12     # 4 cycles per wave64 FP64 instruction per SIMD
    timed = {"fwd": {"ms_per_launch": 0.0768}, "inv": {"ms_per_launch": 0.0765}}
    b = bench.ntt_roofline_block(timed, d, cus=256, timed_sclk_mhz=2100.0)
    assert abs(b["fwd"]["achieved"] - 1024 * 262144 / 76.8e-6 / 1e9) < 1e-6 and abs(b["fwd"]["frac"] - b["fwd"]["achieved"] / 8000.0) < 1e-12
    assert abs(b["fwd"]["alu"]["issue_us_per_launch"] - (2.3e7 + 1.0e5) * 4 / 1024 / 2100.0) < 1e-9
    assert abs(b["fwd"]["alu"]["achieved_frac"] - b["fwd"]["alu"]["issue_us_per_launch"] / 76.8) < 1e-12
    assert b["fwd"]["traffic"] == f["traffic_bytes_per_launch"] and b["inv"]["alu"]["shader_clock_source"].startswith("hwmon")
    assert bench.ntt_roofline_block(timed, None, 256)["fwd"]["traffic"] is None              # --no-pmc: the HBM fraction alone
