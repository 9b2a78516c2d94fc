"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the drop-in classes build the reference's topology / state_dict layout, permutation layers fold
into the right plane maps, and the product path refuses to run without a GPU (no silent fallback)."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from nf_amd import _lib
from nf_amd.normalizing_flows.manager import PWLinManager, PWQuadManager, get_bin
from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad
from nf_amd.normalizing_flows.sequential import FlowSequential

HAS_GPU = torch.cuda.is_available()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "nis_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nis_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = header_symbols()
    assert set(_lib.EXPORTS) <= set(syms)
    for s in syms:
        assert hasattr(L, s), f"libnis_hip.so does not export {s}"
    assert L.nis_version() >= 100


def test_struct_layout_matches_header():
    # sizes the C side computes from the same descriptor must agree with the parameter count of the modules
    L = _lib.lib()
    for cls, kind in ((PWQuad, "pwquad"), (PWLin, "pwlin")):
        cell = cls(8, 3, 6, [7, 5])
        from nf_amd.flow import cell_desc, _param_tensors
        d = cell_desc(cell, list(range(8)))
        ps, rs, bns = _param_tensors(cell)
        assert L.nis_cell_param_count(ctypes.byref(d)) == sum(p.numel() for p in ps)
        assert L.nis_cell_running_count(ctypes.byref(d)) == sum(getattr(bn, n).numel() for bn, n in rs)
        assert sum(p.numel() for p in ps) == sum(p.numel() for p in cell.parameters())


def build_manager(m):
    """create_model without a GPU: the model object is assigned before the warm-up forward raises."""
    torch.manual_seed(0)
    if m["kind"] == "pwquad":
        NF = PWQuadManager(n_flow=m["d"])
        args = (m["n_cells"], m["K"], list(m["NN"]))
    else:
        NF = PWLinManager(n_flow=m["d"])
        args = (m["P"], m["n_cells"], m["K"], list(m["NN"]), m["roll"])
    if HAS_GPU:
        NF.create_model(*args)
    else:
        with pytest.raises(_lib.NisError):
            NF.create_model(*args)
    return NF


@pytest.mark.parametrize("case", golden("models.npz").cases())
def test_topology_and_state_dict_layout(case, g_models):
    m = g_models.meta[case]
    NF = build_manager(m)
    model = NF._model
    assert isinstance(model, FlowSequential)
    names = [n + ":" + type(c).__name__ for n, c in model.named_children()]
    assert names == m["children"]
    assert [c.pass_through_size for c in model.children() if hasattr(c, "KIND")] == m["passes"]
    sd_ref = g_models.sub(case, "sd")
    sd = model.state_dict()
    assert list(sd.keys()) == list(sd_ref.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(sd_ref[k].shape), k
    # loading the reference's state_dict works unchanged
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd_ref.items()})


@pytest.mark.parametrize("case", golden("models.npz").cases())
def test_plane_map_equals_running_the_permutation_layers(case, g_models):
    """Folding Roll/Mask/DeMask into plane indices must equal executing them: push a tensor whose columns
    hold their own index through the permutation layers (cells skipped: they keep column positions)."""
    m = g_models.meta[case]
    model = build_manager(m)._model
    cells, in_cols, out_col, d = model.plan()
    x = torch.arange(d + 1, dtype=torch.double).repeat(3, 1)
    seen = []
    for mod in model.children():
        if hasattr(mod, "KIND"):
            seen.append([int(v) for v in x[0, :d]])
        else:
            x = mod(x)
    assert seen == in_cols
    assert [int(v) for v in x[0, :d]] == out_col
    assert int(x[0, d]) == d


def test_get_bin():
    assert get_bin(5, 4) == [0, 1, 0, 1]
    assert get_bin(0) == [0]


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU behaviour")
def test_no_silent_cpu_fallback():
    cell = PWQuad(2, 1, 4, [3, 3, 3]).double()
    with pytest.raises(_lib.NisError):
        cell(torch.rand(8, 3, dtype=torch.double))
    from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace
    gen = FlatInvertiblePhasespace([100., 100.], [100.] * 4)
    with pytest.raises(_lib.NisError):
        gen.generateKinematics_batch(1000., torch.rand(8, 8, dtype=torch.double))
    from nf_amd.flow import moments
    with pytest.raises(_lib.NisError):
        moments(torch.rand(10, dtype=torch.double))


def test_product_never_imports_the_oracle():
    bad = []
    for base in ("nf_amd", "nisrep"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_affine_cell_param_count_matches_abi():
    from nf_amd.flow import cell_desc, _param_tensors
    from nf_amd.normalizing_flows.layers.coupling_cells import AffineCoupling
    cell = AffineCoupling(7, 3, [6, 5])
    d = cell_desc(cell, list(range(7)))
    ps, rs, _ = _param_tensors(cell)
    L = _lib.lib()
    assert d.kind == _lib.NIS_AFFINE and L.nis_cell_param_count(ctypes.byref(d)) == sum(p.numel() for p in ps) == \
        sum(p.numel() for p in cell.parameters())
    assert [tuple(p.shape) for p in ps] == [tuple(p.shape) for p in cell.parameters()]      # blob order == Sequential order
    assert L.nis_cell_running_count(ctypes.byref(d)) == sum(getattr(bn, n).numel() for bn, n in rs)


def test_experiment_driver_import_path():
    from nisrep.utils.experiment_mg import epoch_length, pro
    assert callable(pro) and epoch_length == 800


def test_nisrep_import_paths():
    import importlib
    mgr = importlib.import_module("nisrep.normalizing_flows.manager")
    assert mgr.PWQuadManager is PWQuadManager and hasattr(mgr, "torch") and hasattr(mgr, "normal")
    ps = importlib.import_module("nisrep.PhaseSpace.flat_phase_space_generator")
    assert hasattr(ps, "FlatInvertiblePhasespace") and hasattr(ps, "PhaseSpaceGeneratorError")
    gen = ps.FlatInvertiblePhasespace([0., 0.], [0.] * 4)
    assert gen.nDimPhaseSpace() == 8
    np.testing.assert_allclose(ps.FlatInvertiblePhasespace.get_flatWeights(1000., 4),
                               golden("rambo.npz").meta["m2to4_massive"]["flat"], rtol=1e-14)


def test_reference_checkpoints_load_and_training_state_round_trips(tmp_path, g_models):
    """the reference's `torch.save` dicts (manager.py:107-109, :360-369) load unchanged; the
    added resume state (model, best model, optimizer, bookkeeping, epoch) survives a save / load cycle."""
    case = golden("models.npz").cases()[0]
    m = g_models.meta[case]
    sd_ref = {k: torch.as_tensor(v) for k, v in g_models.sub(case, "sd").items()}
    # what the reference writes at the end of training, and its initial dump
    full = {"best_epoch": 17, "best_loss": torch.tensor(0.123), "int_loss": torch.tensor(0.5), "best_loss_rel": torch.tensor(0.246),
            "best_func_count": 123456, "model_state_dict": sd_ref, "integ": torch.tensor(0.2322), "err": torch.tensor(1e-4)}
    torch.save(full, tmp_path / "torch")
    torch.save({"model_state_dict": sd_ref}, tmp_path / "torch_int")
    NF = build_manager(m)
    NF.load_checkpoint(str(tmp_path / "torch_int"))
    for k, v in NF._model.state_dict().items():
        np.testing.assert_allclose(v.double().cpu().numpy(), sd_ref[k].double().numpy(), rtol=1e-6, err_msg=k)
    NF2 = build_manager(m)
    ck = NF2.load_checkpoint(str(tmp_path / "torch"))
    assert NF2.best_epoch == 17 and NF2.best_func_count == 123456 and float(NF2.integ_tot) == float(full["integ"])
    assert NF2.best_model is NF2._model and set(ck) == set(full)
    # resume state
    opt = torch.optim.Adamax(NF2._model.parameters(), lr=2e-3, weight_decay=1e-4)
    for p in NF2._model.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    NF2.history = [torch.tensor(1.0), torch.tensor(0.5)]
    NF2.best_var = 0.25
    NF2.save_training_state(str(tmp_path / "resume"), opt, epoch=2)
    NF3 = build_manager(m)
    opt3 = torch.optim.Adamax(NF3._model.parameters(), lr=2e-3, weight_decay=1e-4)
    assert NF3.load_training_state(str(tmp_path / "resume"), opt3) == 2
    for (k, a), (_, b) in zip(NF2._model.state_dict().items(), NF3._model.state_dict().items()):
        assert torch.equal(a, b), k
    sa, sb = opt.state_dict()["state"], opt3.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) for i in sa)
    assert NF3.best_epoch == 17 and NF3.best_var == 0.25 and len(NF3.history) == 2
    with pytest.raises(AttributeError, match="No model was instantiated"):
        type(NF)(n_flow=2).load_checkpoint(str(tmp_path / "torch"))


def _rambo_object():
    obj = os.path.join(ROOT, "nf_amd", "csrc", "build", "rambo.o")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(obj) or not os.path.exists(os.path.join(llvm, "llvm-readelf")):
        pytest.skip("no built rambo.o / llvm tools here")
    return obj, llvm


def test_pipelined_rambo_instantiations_do_not_spill():
    """k_rambo's hand-pipelined row loads (inline-asm issue, hand-written s_waitcnt) are sound only while the compiler never spills the
    registers they land in: the path is compiled for maps of at most six final-state legs (rambo.hip: PIPE_OK), and none of those
    instantiations may report a spilled vector register or any scratch in the built object's metadata."""
    import re
    import shutil
    import subprocess
    import tempfile
    obj, llvm = _rambo_object()
    td = tempfile.mkdtemp()
    try:
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", f".hip_fatbin={td}/fat.bin", obj], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={td}/fat.bin",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={td}/dev.co"], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", f"{td}/dev.co"], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(td, ignore_errors=True)
    seen = 0
    for blk in notes.split(".agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
        sspill = re.search(r"\.sgpr_spill_count:\s+(\d+)", blk)
        scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        if not name or not spill or not sspill or not scratch:
            continue
        m = re.match(r"_Z7k_ramboILi(\d+)ELi(\d+)E", name.group(1))
        if m and int(m.group(1)) <= 6:
            seen += 1
            what = f"k_rambo<{m.group(1)}, {m.group(2)}>"
            assert int(spill.group(1)) == 0, f"{what} spills {spill.group(1)} vector registers"
            # (scalar spills -- up to 95 in the PDF / cut modes -- go to lanes of VGPRs the compiler reserves for them, never to memory here;
            # whether ANY instruction, v_writelane included, touches a row-image register in flight is what the ISA walk below checks)
            assert int(scratch.group(1)) == 0, f"{what} uses {scratch.group(1)} bytes of scratch"
    assert seen >= 20, seen          # n = 2 .. 6, four modes each


def test_pipelined_rambo_loads_are_never_touched_in_flight():
    """The check the spill count cannot give: walk the built gfx950 code of every k_rambo instantiation with the
    hardware's VM_CNT rule (tests/isa_vmem_check.py) -- a register with an outstanding vector-memory load must not be named by any
    instruction until an `s_waitcnt vmcnt(n)` with enough younger operations behind the load has passed, on EVERY path (loop back-edge
    included).  For the compiler's own loads that holds by construction; for the inline-asm `global_load_dwordx4` of the pipelined
    path it proves that the hand-written wait is sufficient and that no copy / re-allocation touches the row image between issue and
    wait.  The checker itself is tested: weakening the hand-written wait by one must produce violations."""
    import re
    import shutil
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import isa_vmem_check as chk
    obj, _ = _rambo_object()
    td = tempfile.mkdtemp()
    try:
        text = chk.disassemble(obj, td)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    fns = chk.functions(text)
    seen = pipelined = 0
    for name, (_, insts) in fns.items():
        m = re.match(r"_Z7k_ramboILi(\d+)ELi(\d+)E", name)
        if not m:
            continue
        nf, mode = int(m.group(1)), int(m.group(2))
        if any(mn.startswith("s_setpc") for _, mn, _, _ in insts):        # (2 -> 10 with cuts calls out of line: no pipelined loads there)
            assert nf > 6, name
            continue
        viol, stats = chk.check_function(insts, name)
        assert not viol, f"k_rambo<{nf}, {mode}>: " + "; ".join(f"{a:#x} {t} touches {r}" for a, t, r in viol[:5])
        assert stats["loads"] > 0 and stats["waits"] > 0
        seen += 1
        if nf <= 6 and nf % 2 == 0:                                        # even row length: the 16-byte pipelined loads exist
            lm_half = 2 * (nf + 2)
            hand = [i for i, (_, mn, ops, _) in enumerate(insts) if mn == "s_waitcnt" and ops.strip() == f"vmcnt({lm_half})"]
            assert hand, f"k_rambo<{nf}, {mode}>: the hand-written s_waitcnt vmcnt({lm_half}) is not in the code"
            assert sum(1 for _, mn, _, _ in insts if mn == "global_load_dwordx4") >= 2 * (3 * nf - 4 + (2 if mode >= 1 else 0)) // 2
            pipelined += 1
            # self-test of the checker on the plain map (in the PDF / cut modes the compiler's own waits for its table loads cover the row
            # image as well): a hand-written wait that tolerates more outstanding operations than the tile has stores must be flagged
            if mode == 0:
                bad = list(insts)
                for i in hand:
                    a, mn, _, t = bad[i]
                    bad[i] = (a, mn, f"vmcnt({lm_half + 2})", t)
                v2, _ = chk.check_function(bad, name)
                assert v2, f"k_rambo<{nf}, {mode}>: weakened wait not detected"
    assert seen >= 30 and pipelined >= 12, (seen, pipelined)


def test_relative_image_row_map_covers_every_row_once(tmp_path):
    """nis_rel_row (nf_amd/csrc/nis_common.h): the relative last-layer image of a PWQuad cell gives a column 2K rows -- K heights minus the
    row of V[K], K - 1 widths minus the row of W[K-1], one zero row.  Host-compiled check of the map the packer uses: every reference row
    except the two subtracted ones appears exactly once, with the right row subtracted; padding rows map to -1."""
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    src = tmp_path / "relmap.cpp"
    src.write_text(r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include "nis_common.h"
int main() {
  for (int K : {4, 8, 16, 32})
    for (int T = 1; T <= 9; ++T) {
      const int o = nis_rel_out(K), R = T * (2 * K + 1);
      int seen[9 * 65] = {0};
      for (int j = 0; j < 16 * ((T * o + 15) / 16); ++j) {
        int js = -2;
        const int jo = nis_rel_row(T, K, j, &js);
        const int t = j / o, q = j % o;
        if (t >= T || q == o - 1) { if (jo != -1 || js != -1) { printf("pad K=%d T=%d j=%d\n", K, T, j); return 1; } continue; }
        if (jo < 0 || jo >= R || js < 0 || js >= R) { printf("range K=%d T=%d j=%d\n", K, T, j); return 1; }
        const int b = t * (2 * K + 1);
        const bool height = q < K;
        if (js != (height ? b + K : b + 2 * K) || jo != (height ? b + q : b + K + 1 + (q - K))) { printf("map K=%d T=%d j=%d\n", K, T, j); return 1; }
        ++seen[jo];
      }
      for (int r = 0; r < R; ++r) {
        const int q = r % (2 * K + 1);
        const int want = (q == K || q == 2 * K) ? 0 : 1;
        if (seen[r] != want) { printf("cover K=%d T=%d r=%d seen=%d\n", K, T, r, seen[r]); return 1; }
      }
    }
  printf("ok\n");
  return 0;
}
''')
    exe = tmp_path / "relmap"
    r = subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "nf_amd", "csrc"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_library_is_loaded_behind_torch():
    """nf_amd._lib.lib() must import torch before it opens libnis_hip.so: torch's wheel carries its own HIP runtime, and the library has to
    bind to that one (loaded first it brought /opt/rocm's copy along: two runtimes in one process, every launch failing with "no
    ROCm-capable device" -- __graft_entry__.build() followed by smoke() in one process)."""
    import subprocess
    import sys
    code = ("import sys; from nf_amd import _lib; assert 'torch' not in sys.modules; _lib.lib(); assert 'torch' in sys.modules; "
            "hip = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}); print(len(hip), hip)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.split()[0] == "1", r.stdout          # exactly one HIP runtime mapped
