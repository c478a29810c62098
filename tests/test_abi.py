"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the
entry points include/nsvd.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "nsvd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nsvd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_binding_binds():
    from neural_svd_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol():
    from neural_svd_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    typed = _lib.load()
    assert typed.nsvd_abi_version() == _lib.ABI_VERSION


def test_host_side_queries_work_without_gpu():
    from neural_svd_amd import hip_ops as H
    shape = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128))
    n = H.workspace_bytes(shape, 512)
    assert n > 0 and n % 256 == 0
    assert H.path_name(shape, 512, H.PATH_GENERIC) == "generic"
    bad = H.ModelShape(L=16, D=2, m=1024, hidden=(128, 7))  # fine: any widths
    assert H.workspace_bytes(bad, 8) > 0


def test_struct_layout_matches_header():
    """sizeof (and two field offsets) of the PODs as the C compiler sees them == ctypes' view."""
    import subprocess
    import tempfile
    from neural_svd_amd import _lib
    prog = r'''
    #include <stdio.h>
    #include "nsvd.h"
    #include <stddef.h>
    int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(nsvd_model_desc), sizeof(nsvd_params),
                       sizeof(nsvd_problem), sizeof(nsvd_tower_params), sizeof(nsvd_rmsprop), sizeof(nsvd_cdk_step_desc),
                       offsetof(nsvd_cdk_step_desc, lr), offsetof(nsvd_cdk_step_desc, gemm_bf16),
                       offsetof(nsvd_rmsprop, state), sizeof(nsvd_step_state), offsetof(nsvd_step_state, cur),
                       offsetof(nsvd_step_state, ema_decay), offsetof(nsvd_cdk_step_desc, grad_scaler),
                       sizeof(nsvd_grad_scaler), offsetof(nsvd_grad_scaler, steps_ok)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    assert [int(v) for v in out] == [ctypes.sizeof(_lib.ModelDesc), ctypes.sizeof(_lib.Params),
                                     ctypes.sizeof(_lib.Problem), ctypes.sizeof(_lib.TowerParams),
                                     ctypes.sizeof(_lib.Rmsprop), ctypes.sizeof(_lib.CdkStepDesc),
                                     _lib.CdkStepDesc.lr.offset, _lib.CdkStepDesc.gemm_bf16.offset,
                                     _lib.Rmsprop.state.offset, ctypes.sizeof(_lib.StepState), _lib.StepState.cur.offset,
                                     _lib.StepState.ema_decay.offset, _lib.CdkStepDesc.grad_scaler.offset,
                                     ctypes.sizeof(_lib.GradScalerState), _lib.GradScalerState.steps_ok.offset]


def _torch_ext():
    import importlib.util
    from neural_svd_amd import _lib
    path = os.path.join(ROOT, "neural_svd_amd", "_nsvd_torch.so")
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(path):
        # optional component, built on demand (`make torch_binding`: host g++ against the torch headers)
        import subprocess
        rc = subprocess.call(["make", "-C", os.path.join(ROOT, "neural_svd_amd", "csrc"), "torch_binding"])
        if rc != 0 or not os.path.exists(path):
            pytest.skip("the optional torch binding does not build with this toolchain")
    _lib.load()
    spec = importlib.util.spec_from_file_location("_nsvd_torch", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_torch_binding_agrees_with_the_header():
    """the tensor-level binding (csrc/torch_binding.cpp, NSVD_BINDING=torch): built, loadable, on the same ABI version,
    its compiler's view of the structs == ctypes' == the header's, and every entry point it binds is one the header
    declares"""
    from neural_svd_amd import _lib
    tb = _torch_ext()
    assert tb.abi_version() == _lib.ABI_VERSION
    assert tb.struct_layout() == [ctypes.sizeof(_lib.ModelDesc), ctypes.sizeof(_lib.Params), ctypes.sizeof(_lib.Problem),
                                  ctypes.sizeof(_lib.Rmsprop), _lib.Rmsprop.state.offset, ctypes.sizeof(_lib.StepState),
                                  _lib.StepState.cur.offset]
    bound = tb.bound_entry_points()
    assert set(bound) <= set(_declared()) and "nsvd_operator_forward" in bound and len(bound) >= 12
    # host-side queries work without a GPU, tensors are refused in C++
    import torch
    sh = tb.Shape(16, 2, 1024, [128, 128, 128, 1], False)
    from neural_svd_amd import hip_ops as H
    assert tb.workspace_bytes(sh, 512) == H.workspace_bytes(H.ModelShape(L=16, D=2, m=1024, hidden=(128, 128, 128)), 512)
    assert tb.path_name(sh, None, 512, 0) == "fused_mfma"
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        tb.evd_moments(torch.zeros(4, 2), torch.zeros(4, 2), 1, None, torch.zeros(9), torch.zeros(16))


def test_ops_refuse_cpu_tensors():
    import torch
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd._lib import NsvdError
    with pytest.raises(NsvdError):
        H.evd_moments(torch.zeros(4, 2), torch.zeros(4, 2), H.MASK_SEQUENTIAL, None,
                      moments=torch.zeros(9), scratch=torch.zeros(1024, dtype=torch.uint8))
