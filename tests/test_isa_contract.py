"""The machine code of the stream-K kernel (gemm_hls_amd/csrc/mm_mfma_f32_streamk.inc, a part of mm_mfma_f32.hip: mfma_f32_streamk_teams_kernel), checked on the gfx950
ISA hipcc emits for the shipped source with the build's flags.

In the form MM_PATH_AUTO runs (Combine::LastArriver) the parts of a tile that a range boundary cuts go to scratch slots, each
part raises an epoch flag and then LOOKS at its siblings' flags, and the last to arrive adds the slots into C -- across XCDs,
with hand-written agent-scope (sc1) stores.  Nobody waits; but who gathers, and whether the gatherer sees the slots, is a
contract between the source (inline asm included) and the compiler: a compiler bump or an innocent-looking edit can break it
silently -- results would still be right most of the time.  This test pins what a reviewer checks by hand:

  every part   slot stores are `global_store_dwordx4 ... sc1`, each followed by a wait state (`s_nop`: a store of more than 8 bytes
               reads its data registers late, the next VALU write to them would otherwise be stored instead); then
               `s_waitcnt vmcnt(0)` (each wavefront: its slot stores have reached the coherence point), `s_barrier` (all
               wavefronts have), and only then ONE lane's `global_store_dwordx2 ... sc1` of the flag -- no other store in between;
  raise, look  `s_waitcnt vmcnt(0)` between the flag store and the first sibling-flag load, the loads `sc1`: the store-then-load
               order of a sequentially consistent pair on this target, without which two parts finishing together could both
               miss each other and nobody would gather;
  gather       `buffer_inv sc1` (drop stale lines before the slots are read) after the looks, `s_barrier` (the other wavefronts do
               not start early), and only then the gather's `global_load_dwordx4`;
  wait-free    no `s_sleep`, no atomics on data (the order of additions is fixed: kernel/Compute.cpp:108-142, one deterministic
               k-ordered result per element), no whole-L2 write-back.

The cross-check form (Combine::FixupKernel + streamk_teams_fixup_kernel, f32_splitk 11) must contain no inter-workgroup
communication at all; and no kernel of the translation unit may spill.

CPU test: hipcc cross-compiles the translation unit to assembly here (about 15 s)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gemm_hls_amd", "csrc", "mm_mfma_f32.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
_ALL = {}


@pytest.fixture(scope="module")
def ordered_kernels():
    from gemm_hls_amd import build
    flags = [f for f in build.COMMON if f != "--offload-compress"]          # the flags the shipped object is built with
    r = subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", SRC, "-o", "-"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.split("\n")
    kernels, name, body = {}, None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*; @", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = ln.split(";")[0].strip()
        if ln.lstrip().startswith(";;#ASM"):        # keep inline-asm brackets out, their contents in
            continue
        if t and not t.startswith("."):
            body.append(t)
        if "s_endpgm" in t:
            kernels[name] = body
            name = None
    picked = {k: v for k, v in kernels.items() if "mfma_f32_streamk_teams_kernel" in k}
    assert len(picked) == 6, list(kernels)           # (scalar-base DMA, vector-address) x Combine::{FixupKernel, LastArriver, Ticket}
    _ALL["kernels"] = kernels
    _ALL["metadata"] = r.stdout
    return _by_combine(1)


def _by_combine(value):
    """The two instantiations (DMA forms) of mfma_f32_streamk_teams_kernel<G, Combine(value)>: 0 FixupKernel, 1 LastArriver, 2 Ticket."""
    picked = {k: v for k, v in _ALL["kernels"].items() if "mfma_f32_streamk_teams_kernel" in k and f"CombineE{value}EEEv" in k}
    assert len(picked) == 2, [k for k in _ALL["kernels"] if "streamk_teams" in k]
    return picked


def _is(op, ins):
    return ins.split()[0] == op


def test_slot_stores_are_agent_scope_and_followed_by_a_wait_state(ordered_kernels):
    for name, body in ordered_kernels.items():
        idx = [i for i, t in enumerate(body) if _is("global_store_dwordx4", t) and t.endswith("sc1")]
        assert len(idx) >= 16, (name, len(idx))      # a 128 x 128 tile of a 256-thread workgroup: 16 quads per thread, per write-back form
        for i in idx:
            assert _is("s_nop", body[i + 1]), (name, body[i], body[i + 1])


def test_no_atomics_on_data_and_no_whole_cache_write_back(ordered_kernels):
    for name, body in ordered_kernels.items():
        assert not any("atomic" in t for t in body), name
        assert not any(_is("buffer_wbl2", t) for t in body), name      # slots go out with sc1 stores, not by writing the L2 back


def test_the_two_kernel_cross_check_form_has_no_inter_workgroup_communication(ordered_kernels):
    """mfma_f32_streamk_teams_kernel<G, Combine::FixupKernel> + streamk_teams_fixup_kernel (f32_splitk 11, the cross-check of the
    default): no inter-workgroup communication at all -- no flag store, no flag load, no sleep, no invalidate, no agent-scope store."""
    for name, body in _by_combine(0).items():
        assert not any(_is("global_store_dwordx2", t) for t in body), name
        assert not any(_is("global_load_dwordx2", t) and t.endswith("sc1") for t in body), name
        assert not any(_is("s_sleep", t) for t in body), name
        assert not any(t.startswith("buffer_inv") for t in body), name
        assert not any(t.startswith("global_store") and t.endswith("sc1") for t in body), name
        assert not any("atomic" in t for t in body), name
    fix = [k for k in _ALL["kernels"] if "streamk_teams_fixup_kernel" in k]
    assert len(fix) == 1 and not any("atomic" in t or _is("s_sleep", t) for t in _ALL["kernels"][fix[0]])


def test_the_counter_ticket_cross_check_form_is_what_the_language_model_compiles_to(ordered_kernels):
    """mfma_f32_streamk_teams_kernel<G, Combine::Ticket> (f32_splitk 12, round 6): the canonical last-block pattern -- plain slot
    stores, an agent-scope RELEASE fence in every thread (on this target: the L2 write-back the shipped form avoids, which is what
    it costs), ONE compare-exchange per part on the tile's counter, an agent-scope ACQUIRE (invalidate) before the gather; no
    hand-placed sc1 stores, no flag loads, no sleep.  Pinned so that the cross-check stays the independent implementation it is."""
    for name, body in _by_combine(2).items():
        assert any(_is("buffer_wbl2", t) for t in body), name                                   # the release fence
        assert sum(1 for t in body if t.startswith("global_atomic_cmpswap")) >= 1, name          # the ticket
        assert any(t.startswith("buffer_inv") for t in body), name                              # the acquire
        assert not any(t.startswith("global_store_dwordx4") and t.endswith("sc1") for t in body), name
        assert not any(_is("s_sleep", t) for t in body), name


def test_no_shipped_matrix_core_kernel_of_this_unit_spills(ordered_kernels):
    """No scratch in any kernel of mm_mfma_f32.hip (the VectorAddress twin of the 256 x 256 geometry once
    kept 8 bytes of its addresses in scratch): no scratch_ instruction, and a zero private segment in the metadata."""
    for name, body in _ALL["kernels"].items():
        assert not any(t.startswith("scratch_") for t in body), name
    sizes = re.findall(r"\.private_segment_fixed_size:\s*(\d+)", _ALL["metadata"])
    assert sizes and all(int(x) == 0 for x in sizes), sizes


def test_the_last_arriver_form_is_wait_free_and_orders_raise_before_look(ordered_kernels):
    """mfma_f32_streamk_teams_kernel<G, Combine::LastArriver>, what MM_PATH_AUTO runs: every part stores its slot at agent scope
    (sc1 + wait state), then s_waitcnt vmcnt(0) -> s_barrier -> ONE sc1 flag store -> s_waitcnt vmcnt(0) -> sc1 flag LOADS (the
    store-then-load order of a sequentially consistent pair on this target) -> buffer_inv sc1 -> s_barrier -> the gather's loads.
    And it never waits: no s_sleep, and no flag load sits in a loop that polls (the only backward branches after the flag store
    are the loop over the tile's parts and the segment loop, both of which advance)."""
    for name, body in _by_combine(1).items():
        slot_stores = [i for i, t in enumerate(body) if _is("global_store_dwordx4", t) and t.endswith("sc1")]
        assert len(slot_stores) >= 16, (name, len(slot_stores))
        for i in slot_stores:
            assert _is("s_nop", body[i + 1]), (name, body[i], body[i + 1])
        flag = [i for i, t in enumerate(body) if _is("global_store_dwordx2", t)]
        assert len(flag) == 1 and body[flag[0]].endswith("sc1"), (name, [body[i] for i in flag])
        i = flag[0]
        back = body[max(0, i - 40):i]
        bar = max(j for j, t in enumerate(back) if _is("s_barrier", t))
        assert not any(t.startswith("global_store") or t.startswith("buffer_store") for t in back[bar:]), (name, back[bar:])
        assert any(_is("s_waitcnt", t) and "vmcnt(0)" in t for t in back[:bar]), (name, back)
        after = body[i + 1:]
        looks = [j for j, t in enumerate(after) if _is("global_load_dwordx2", t)]
        assert looks and all(after[j].endswith("sc1") for j in looks), (name, [after[j] for j in looks])
        assert any(_is("s_waitcnt", t) and "vmcnt(0)" in t for t in after[:looks[0]]), (name, after[:looks[0]])   # raise completes before the first look
        inv = [j for j, t in enumerate(after) if t == "buffer_inv sc1"]
        assert len(inv) == 1 and inv[0] > looks[0], (name, inv, looks)
        rest = after[inv[0] + 1:]
        first_barrier = next(j for j, t in enumerate(rest) if _is("s_barrier", t))
        first_gather = next(j for j, t in enumerate(rest) if _is("global_load_dwordx4", t))
        assert first_barrier < first_gather, (name, first_barrier, first_gather)
        assert not any(_is("s_sleep", t) for t in body), name                      # nobody waits
        assert not any("atomic" in t for t in body) and not any(_is("buffer_wbl2", t) for t in body), name


# ---- the k-ordered tile kernels (gemm_hls_amd/csrc/mm_valu_tile_fp_exact.hip) ------------------------------------------------------
def test_the_k_ordered_tile_unit_has_no_fused_multiply_add_with_or_without_the_compile_flag():
    """`ordered_tile` must multiply and add as two separately rounded instructions in Data_t -- the reference's arithmetic
    (kernel/Compute.cpp:129-133; binary16 accumulating in binary16) -- or its bits are not Naive's.  The unit says so twice: the
    build passes -ffp-contract=off AND the file carries `#pragma clang fp contract(off)`; this compiles it WITHOUT the flag and
    looks at the machine code of all of its kernels: no fused multiply-add of any floating type, packed or not, no dot
    product; and the packed binary16 pair the half kernels are priced against is what the inner loop is made of."""
    from gemm_hls_amd import build
    src = os.path.join(ROOT, "gemm_hls_amd", "csrc", "mm_valu_tile_fp_exact.hip")
    assert build.EXTRA.get("mm_valu_tile_fp_exact.hip") == ["-ffp-contract=off"] and build.EXTRA.get("mm_ordered.hip") == ["-ffp-contract=off"]
    flags = [f for f in build.COMMON if f != "--offload-compress"]           # NOT the per-file extra flag: the pragma alone must do
    r = subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", src, "-o", "-"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ops = [ln.split()[0] for ln in r.stdout.split("\n") if ln.startswith("\tv_")]
    fused = sorted({op for op in ops if re.match(r"v_(pk_)?(fma|fmac|mad|mac|dot\d*c?)_", op) and not re.search(r"_(u|i)(8|16|24|32|64)", op)})
    assert fused == [], fused
    assert ops.count("v_pk_mul_f16") >= 128 and ops.count("v_pk_add_f16") >= 128           # half: two elements per lane and instruction
    assert ops.count("v_mul_f64_e32") + ops.count("v_mul_f64") >= 64 and ops.count("v_add_f64_e32") + ops.count("v_add_f64") >= 64
    # std::min / std::max to the letter: compare-and-select, not the hardware minNum / maxNum of the fast family
    assert not any(op.startswith(("v_min_f", "v_max_f", "v_min3_f", "v_max3_f", "v_pk_min_f", "v_pk_max_f")) for op in ops), \
        sorted({op for op in ops if op.startswith(("v_min", "v_max", "v_pk_min", "v_pk_max"))})
