"""ISA gate of the hot kernel (CPU, runs on the built library): the MFMAs of conv_zreg_kernel.h are inline asm with AGPR
operands, which hipcc's hazard recogniser does not see - a compiler change could put a VALU write of an MFMA operand right in
front of the MFMA that reads it.  profiles/tools/hazard_scan.py disassembles every conv3_zreg_kernel instantiation of
libdelivr_hip.so and this test fails loudly on any unpadded VALU-write -> MFMA-read pair, on scratch use, on a register
budget that no longer fits one wave per SIMD, or on an unexpected accumulator-register count."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "delivr_cfos_amd", "lib", "libdelivr_hip.so")


def _tool():
    spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(ROOT, "profiles", "tools", "hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _need_tools():
    import pytest

    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.isfile(os.path.join(llvm, "llvm-readelf")) and os.path.isfile(os.path.join(llvm, "llvm-objdump"))):
        pytest.skip("ROCm LLVM tools (llvm-readelf / llvm-objdump) are not installed on this host")
    if not os.path.isfile(LIB):  # a missing library is built (as tests/test_abi_cpu.py does), never skipped: the gate stays on
        import __graft_entry__ as g

        g.build()
    assert os.path.isfile(LIB), "libdelivr_hip.so is not built (make -C delivr_cfos_amd/csrc)"


def test_scanner_finds_a_planted_readback_hazard():
    hs = _tool()
    mf = "v_mfma_f32_16x16x32_f16 a[0:3], a[10:13], v[4:7], a[0:3]"
    planted = [mf, "v_add_u32 v9, v9, v9", "v_accvgpr_read_b32 v20, a1"]
    padded = [mf, "s_nop 7", "s_nop 3", "v_accvgpr_read_b32 v20, a1"]
    other = [mf, "v_accvgpr_read_b32 v20, a9"]
    chained = [mf, "v_mfma_f32_16x16x32_f16 a[0:3], a[14:17], v[4:7], a[0:3]"]
    assert len(hs.scan_readback(planted)) == 1 and hs.scan_readback(planted)[0][4] == 1
    assert hs.scan_readback(padded) == [] and hs.scan_readback(other) == [] and hs.scan_readback(chained) == []
    assert len(hs.scan_readback(["v_mfma_f32_16x16x32_f16 v[0:3], a[10:13], v[4:7], v[0:3]", "buffer_store_dwordx4 v[0:3], v9, s[0:3], 0 offen"])) == 1


def test_scanner_finds_a_planted_hazard():
    hs = _tool()
    clean = ["v_mov_b32 v5, v1", "s_nop 1", "v_mfma_f32_16x16x32_f16 a[0:3], a[10:13], v[4:7], a[0:3]"]
    planted = ["v_mov_b32 v5, v1", "v_mfma_f32_16x16x32_f16 a[0:3], a[10:13], v[4:7], a[0:3]"]
    far = ["v_mov_b32 v5, v1", "v_add_u32 v9, v9, v9", "v_add_u32 v8, v8, v8", "v_mfma_f32_16x16x32_f16 a[0:3], a[10:13], v[4:7], a[0:3]"]
    assert hs.scan(clean) == ([], 1)
    found, n = hs.scan(planted)
    assert n == 1 and len(found) == 1 and found[0][4] == 0
    assert hs.scan(far)[0] == []


# (Cin, tile rows, activate-on-load) -> AGPRs that hold the weight fragments + what the schedule pins
EXPECTED_MFMA = {("Li32E", "Li16E"): 2592, ("Li32E", "Li8E"): 1296, ("Li64E", "Li8E"): 2592}


def test_zreg_kernels_have_no_mfma_hazard_no_scratch_and_fit_one_wave_per_simd():
    _need_tools()
    hs = _tool()
    rep = hs.library_report(LIB)
    assert len(rep) == 20, sorted(rep)  # 2 formats x (Cin 32: t8 / t16 x a0 / a1, t8 / t16 with an addend x a0 / a1; Cin 64: t8 a0/a1)
    for name, r in rep.items():
        assert r["hazards"] == 0, (name, r["first"])
        assert r["readback_hazards"] == 0, (name, r["first_readback"])  # asm MFMA result read too early by non-MFMA code
        assert r.get("private_segment_fixed_size", 0) == 0, (name, "scratch")
        assert r.get("vgpr_spill_count", 0) == 0, name
        # unified register file: 512 per lane and SIMD; the metadata's vgpr_count is the unified total (arch VGPRs rounded up
        # to the AGPR base + AGPRs): one wave per SIMD must fit
        v, a = r["vgpr_count"], r["agpr_count"]
        assert a < v <= 512, (name, v, a)
        # the weight fragments live in AGPRs: 27 taps x Cin/32 fragments x 4 registers, + the pinned accumulators
        cin = 64 if "Li64E" in name else 32
        assert a >= 27 * (cin // 32) * 4, (name, a)
        key = [k for k in EXPECTED_MFMA if k[0] in name and k[1] in name]
        assert key and r["mfma"] == EXPECTED_MFMA[key[0]], (name, r["mfma"])


def test_persistent_upconv_kernel_has_no_mfma_hazard_and_keeps_its_weights_in_agprs():
    """upconv.hip's persistent kernel: asm MFMAs like the z-reg conv, 32 A-fragments (128 AGPRs) per wave for the whole walk over
    the tiles, one tile's 32 iterations x 32 MFMAs unrolled, the halo tile staged by LDS-DMA (no scratch, no spills)."""
    _need_tools()
    hs = _tool()
    rep = hs.library_report(LIB, name_filter=("upconv2m_kernel",))
    assert len(rep) == 2, sorted(rep)  # fp16, bf16
    for name, r in rep.items():
        assert r["hazards"] == 0 and r["readback_hazards"] == 0, (name, r["first"], r["first_readback"])
        assert r.get("private_segment_fixed_size", 0) == 0 and r.get("vgpr_spill_count", 0) == 0, name
        assert r["agpr_count"] == 128 and r["vgpr_count"] <= 512, (name, r["vgpr_count"], r["agpr_count"])
        assert r["mfma"] == 32 * 32, (name, r["mfma"])


def test_deep_level_kernels_fit_two_waves_per_simd_without_scratch():
    """conv_deep.hip: 8-wave workgroups = two waves per SIMD, i.e. at most 256 registers per lane, and nothing may spill (a spill
    in the K loop of an MFMA kernel costs more than the kernel gained); compiler-scheduled intrinsic MFMAs (no hand-placed asm
    MFMA: the hazard recogniser sees them), 9 taps x NCB x 4 blocks of them per kz group."""
    _need_tools()
    hs = _tool()
    rep = hs.library_report(LIB, name_filter=("conv3_deep_kernel", "deconv2_deep_kernel"))
    conv = {k: v for k, v in rep.items() if "conv3_deep_kernel" in k}
    dec = {k: v for k, v in rep.items() if "deconv2_deep_kernel" in k}
    assert len(conv) == 8 and len(dec) == 4, sorted(rep)  # 2 formats x (tile 16 / 8 wide) x (64 / 32 channels); 2 formats x Cin 128 / 256
    for name, r in rep.items():
        assert r.get("private_segment_fixed_size", 0) == 0 and r.get("vgpr_spill_count", 0) == 0, (name, "scratch / spills")
        assert r["vgpr_count"] <= 256 and r["agpr_count"] == 0, (name, r["vgpr_count"], r["agpr_count"])
    for name, r in conv.items():
        assert r["mfma"] == (144 if "ELi4EEE" in name else 72), (name, r["mfma"])
    # the LDS-DMA asm writes m0, which hipcc treats as reserved (a clobber on it is not honoured: -Winline-asm).  The kernels are
    # correct as long as m0 has no other user: every mention of m0 is one of our `s_mov_b32 m0, sN`, each LDS-DMA load follows one,
    # and no instruction that reads m0 implicitly (movrel, gws, sendmsg) exists in them
    for name, r in rep.items():
        m0 = r["m0"]
        assert m0["other"] == [], (name, m0["other"][:3])
        assert m0["writes"] >= 1 and m0["dma_reads"] == m0["writes"], (name, m0)
