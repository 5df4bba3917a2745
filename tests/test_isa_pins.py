"""The hand-synchronised kernels against the toolchain (no GPU needed: the gfx950 code objects bundled in csrc/*.o are
disassembled with the ROCm LLVM tools).

csrc/t2s_attn.hip, t2s_rows.h and t2s_rows16.h stream K / V blocks and weight chunks through LDS rings with LDS-DMA
(`global_load_lds_dwordx4` issued from inline asm, hidden from hipcc's waitcnt pass) and wait with HAND-COUNTED
`s_waitcnt vmcnt(N)`: "N younger vector-memory operations may still be in flight, everything older has landed".  That
count is only right while the compiler puts no vector-memory instruction of its own into those loops -- a scratch spill
reload, a global load it sank next to its use, a store it moved -- and a toolchain upgrade could do any of these silently:
the kernel would then read an LDS slot before its DMA has landed, without failing any build step.

Pinned per hot kernel: no scratch (private segment 0, no spills, no scratch_* instruction), VGPRs within the budget of two
waves per SIMD, and for every innermost loop that holds both LDS-DMA and MFMAs the tuple
(MFMAs, LDS-DMAs, other vector loads, vector stores, scratch ops, the vmcnt wait values in program order).
If this test fails after a compiler change: read the new ISA (`python tools/isa_report.py`), re-derive the counts in the
source comments, run `python tools/stress_determinism.py` on a GPU, and only then update the pins."""
import importlib.util
import os
import shutil
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("isa_report", os.path.join(REPO, "tools", "isa_report.py"))
isa = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(isa)

pytestmark = pytest.mark.skipif(not (os.path.exists(os.path.join(isa.LLVM, "llvm-objdump")) and
                                     os.path.exists(os.path.join(REPO, "t2ms_amd", "csrc", "t2s_attn.o"))),
                                reason="needs the ROCm LLVM tools and the built objects (__graft_entry__.build())")

NS = "_ZN3t2s"
# kernel -> (VGPR budget, scratch bytes allowed, [(mfma, lds_dma, loads, stores, scratch ops, vmcnt waits)] streaming loops)
PINS = {
    # attention, 4-slot K / V^T ring (t2s_attn.hip): packed = one head per workgroup pair, persistent = one workgroup per CU
    NS + "22attn_fwd_packed_kernelILi2EEEvPKfS2_S2_Pfi": (256, 0, [(32, 2, 0, 0, 0, (4,)), (64, 2, 0, 0, 0, (4,))]),
    NS + "22attn_fwd_packed_kernelILi4EEEvPKfS2_S2_Pfi": (128, 0, [(32, 2, 0, 0, 0, (4,))]),   # one query tile per wave
    # (round 5: the one-tile wave issues all 8 LDS-DMAs of a block and waits for vmcnt(16); the two-tile waves' loops hold none)
    NS + "26attn_fwd_persistent_kernelEPKfS1_S1_Pfi": (256, 0, [(48, 8, 0, 0, 0, (16,))]),
    # row chain, 3-slot weight ring two chunks ahead (t2s_rows.h): <DO_MLP, DO_QKV>
    NS + "15dit_rows_kernelILb0ELb1EEEvNS_7RowArgsE": (256, 0, [(128, 4, 0, 6, 0, ())]),
    NS + "15dit_rows_kernelILb1ELb1EEEvNS_7RowArgsE": (256, 0, [(128, 8, 0, 0, 0, (4, 4)), (128, 4, 0, 6, 0, ())]),
    NS + "15dit_rows_kernelILb1ELb0EEEvNS_7RowArgsE": (256, 0, [(128, 8, 0, 0, 0, (0, 4, 0))]),
    # the same on 16-token tiles (t2s_rows16.h)
    NS + "17dit_rows16_kernelILb0ELb1EEEvNS_7RowArgsE": (256, 0, [(128, 4, 0, 4, 0, (0,))]),
    NS + "17dit_rows16_kernelILb1ELb1EEEvNS_7RowArgsE": (256, 0, [(128, 8, 0, 0, 0, (4, 4)), (128, 4, 0, 4, 0, (0,))]),
    NS + "17dit_rows16_kernelILb1ELb0EEEvNS_7RowArgsE": (256, 0, [(128, 8, 0, 0, 0, (0, 4, 0))]),
}
# the opt-in bf16x3 row kernels keep a few bytes of scratch: bounded, not pinned loop by loop
X3_SCRATCH_MAX = {NS + "18dit_rows_x3_kernelILb0ELb1EEEvNS_9RowArgsX3E": 0,
                  NS + "18dit_rows_x3_kernelILb1ELb1EEEvNS_9RowArgsX3E": 32,
                  NS + "18dit_rows_x3_kernelILb1ELb0EEEvNS_9RowArgsX3E": 32}


@pytest.fixture(scope="module")
def reports():
    wd = tempfile.mkdtemp(prefix="t2s_isa_")
    try:
        out = {}
        for stem in ("t2s_attn", "t2s_dit"):
            out.update(isa.report(stem, wd))
        yield out
    finally:
        shutil.rmtree(wd, ignore_errors=True)


@pytest.mark.parametrize("kernel", sorted(PINS))
def test_hot_kernel_keeps_its_counted_wait_structure(reports, kernel):
    assert kernel in reports, f"{kernel} is not in the gfx950 code objects (renamed? then rename the pin)"
    r, (vgpr_budget, scratch_max, loops) = reports[kernel], PINS[kernel]
    m = r["meta"]
    name = isa.demangled(kernel)
    assert m["private_segment_fixed_size"] <= scratch_max and m["vgpr_spill_count"] == 0 and m["sgpr_spill_count"] == 0, (name, m)
    assert r["scratch"] == 0, f"{name}: {r['scratch']} scratch_* instructions (every reload is a vmcnt(0) inside a counted ring)"
    assert m["vgpr_count"] <= vgpr_budget, f"{name}: {m['vgpr_count']} VGPRs, two waves per SIMD need <= {vgpr_budget}"
    assert m["wavefront_size"] == 64
    got = isa.streaming_loops(r)
    assert got == loops, (f"{name}: the LDS-DMA loops changed shape\n  pinned {loops}\n  now    {got}\n"
                          "a vector-memory instruction the compiler added (or moved) inside a ring invalidates the hand-counted "
                          "vmcnt waits -- see this file's docstring before touching the pins")
    for mfma, dma, loads, stores, scratch, waits in got:
        assert loads == 0 and scratch == 0, (name, got)


def test_f32_path_uses_the_f32_matrix_instruction(reports):
    """The parity path is fp32 end to end: v_mfma_f32_32x32x2_f32 (16x16x4 on the small-launch tiles), never a reduced-precision form."""
    wd = tempfile.mkdtemp(prefix="t2s_isa_")
    try:
        co = isa.extract_code_object(os.path.join(REPO, "t2ms_amd", "csrc", "t2s_attn.o"), wd)
        dis = isa.disassemble(co)
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    for k in (NS + "22attn_fwd_packed_kernelILi2EEEvPKfS2_S2_Pfi", NS + "22attn_fwd_packed_kernelILi4EEEvPKfS2_S2_Pfi",
              NS + "26attn_fwd_persistent_kernelEPKfS1_S1_Pfi"):
        kinds = {mn for _, mn, _ in dis[k] if mn.startswith("v_mfma")}
        assert kinds == {"v_mfma_f32_32x32x2_f32"}, (k, kinds)


@pytest.mark.parametrize("kernel", sorted(X3_SCRATCH_MAX))
def test_x3_row_kernels_scratch_is_bounded(reports, kernel):
    m = reports[kernel]["meta"]
    assert m["private_segment_fixed_size"] <= X3_SCRATCH_MAX[kernel] and m["vgpr_count"] <= 256, m


# the <proj + MLP ...> row kernels (f32 and bf16x3) leave their residual stream in flight across the FIRST barrier
BUT16 = [NS + "15dit_rows_kernelILb1ELb1EEEvNS_7RowArgsE", NS + "15dit_rows_kernelILb1ELb0EEEvNS_7RowArgsE",
         NS + "17dit_rows16_kernelILb1ELb1EEEvNS_7RowArgsE", NS + "17dit_rows16_kernelILb1ELb0EEEvNS_7RowArgsE",
         NS + "18dit_rows_x3_kernelILb1ELb1EEEvNS_9RowArgsX3E", NS + "18dit_rows_x3_kernelILb1ELb0EEEvNS_9RowArgsX3E"]


@pytest.mark.parametrize("kernel", BUT16)
def test_first_barrier_keeps_only_the_residual_stream_in_flight(kernel):
    """ROWS_SYNC_BUT16 / X3_SYNC_BUT16 (round 5): `s_waitcnt vmcnt(16) lgkmcnt(0)` + `s_barrier` instead of vmcnt(0).  It is
    right only while (a) at least 16 vector-memory operations follow the weight DMAs of the prologue (else a DMA piece could be
    among the 16 youngest and chunk 0 be read before it has landed), (b) the 16 youngest are plain 16-byte loads (the tile's
    residual stream, whose registers the compiler guards itself) -- no store, no scratch, no LDS-DMA -- and (c) the wait
    directly precedes the barrier.  A compiler that moves a load or spills here fails this test, not a GPU run."""
    wd = tempfile.mkdtemp(prefix="t2s_isa_")
    try:
        co = isa.extract_code_object(os.path.join(REPO, "t2ms_amd", "csrc", "t2s_dit.o"), wd)
        insts = isa.disassemble(co)[kernel]
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    idx = next(i for i, (_, mn, ops) in enumerate(insts) if mn == "s_waitcnt" and "vmcnt(16)" in ops and "lgkmcnt(0)" in ops)
    assert insts[idx + 1][1] == "s_barrier", insts[idx:idx + 3]
    assert not any(mn == "s_barrier" for _, mn, _ in insts[:idx]), "the counted wait must belong to the FIRST barrier"
    vm = [(mn, ops) for _, mn, ops in insts[:idx] if isa.VMEM.match(mn)]
    last_dma = max(i for i, (mn, _) in enumerate(vm) if mn.startswith("global_load_lds"))
    assert len(vm) - 1 - last_dma >= 16, f"only {len(vm) - 1 - last_dma} vector-memory operations follow the prologue's weight DMAs"
    youngest = vm[-16:]
    assert all(mn == "global_load_dwordx4" for mn, _ in youngest), youngest
