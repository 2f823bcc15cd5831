"""The C-ABI library loads on a CPU-only box and exports every entry point include/mpcodec.h declares
(no compute calls here); the hot path refuses to run without a device instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "mpcodec.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(mpc_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    import imageexperiments_amd as ia
    lib = ia.load_library()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_library_is_in_tree():
    import imageexperiments_amd as ia
    assert ia.library_path().startswith(ROOT)
    assert os.path.exists(ia.library_path())


def test_no_cpu_fallback_for_the_hot_path():
    """A host-only context exists for the tables and the entropy stage; the tile encoder must refuse it."""
    import imageexperiments_amd as ia
    ctx = ia.create_compression_context(8, 8, 3.5, device=-1)
    with pytest.raises(ia.MpcError) as e:
        ctx.encode_tiles(np.zeros((8, 8, 3), np.uint8))
    assert e.value.status == ia.api.MPC_ERR_NO_DEVICE
    with pytest.raises(ia.MpcError):
        ctx.calc_mp(0, np.zeros((1, 64)))
    with pytest.raises(ia.MpcError):
        ctx.encode_image(np.zeros((8, 8, 3), np.uint8))


def test_argument_errors_are_status_codes():
    import imageexperiments_amd as ia
    for K, bs in ((0, 8), (33, 8), (8, 0), (8, 9)):
        with pytest.raises(ia.MpcError) as e:
            ia.create_compression_context(K, bs, 3.5, device=-1)
        assert e.value.status == ia.api.MPC_ERR_ARGUMENT


def test_product_tables_equal_oracle_tables(oracle):
    """createCompressionContext on the host: dictionary and quant tables bit-identical to the oracle's,
    for several K / quality settings (BASELINE configs sweep 2.0 .. 6.0, K in 8/16/32)."""
    import imageexperiments_amd as ia
    ctx = ia.create_compression_context(32, 8, 3.5, device=-1)
    o = oracle.OracleContext(32, 8, 3.5)
    base, rows, det = ctx.dictionary()
    assert (base.view(np.uint64) == o.base.view(np.uint64)).all()
    assert (rows == o.det_rows).all()
    for ch in range(3):
        assert (det[ch].view(np.uint64) == o.det[ch].view(np.uint64)).all()
    L = oracle.lib()
    for K in (8, 16, 32):
        for bpp in (0.0, 1.0, 2.0, 3.5, 4.5, 6.0, 8.0):
            c = ia.create_compression_context(K, 8, bpp, device=-1) if (K, bpp) != (32, 3.5) else ctx
            q = np.zeros((3, K))
            L.mpo_quant_tables(K, 8, float(bpp), oracle._dp(q[0]), oracle._dp(q[1]), oracle._dp(q[2]))
            assert (c.quant == q).all(), (K, bpp)
            if c is not ctx:
                c.close()


def test_sweep_kernels_keep_scalar_dictionary_operands():
    """Every v_mul_f64 of the sweep kernels in the shipped gfx950 code object reads the dictionary coefficient from
    an SGPR (the s_load-fed design); a silent fallback to vector loads is a 6x slowdown."""
    from imageexperiments_amd.build import verify_scalar_sweeps
    rep = verify_scalar_sweeps()
    if rep is None:
        pytest.skip("llvm-objdump not available")
    assert rep["mp_base_kernel"][1] == 128 and rep["mp_detail_kernel"][1] == 64
    assert all(a == b for a, b in rep.values())
