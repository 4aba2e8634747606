import gzip
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The shared libraries are built in-tree by __graft_entry__.build(); build on demand when a
    fresh checkout runs the CPU suite directly."""
    import __graft_entry__ as g
    import spada_sim_amd._ffi as ffi
    if not os.path.exists(ffi.LIB_PATH):
        g.build()
    from oracle import oracle
    oracle.build()


@pytest.fixture(scope="session")
def matrices_dir(tmp_path_factory):
    """./matrices layout of the reference (config ss_filepath) with cari.mtx unpacked from the fixture."""
    d = tmp_path_factory.mktemp("matrices")
    with gzip.open(os.path.join(GOLDEN, "cari.mtx.gz"), "rb") as f, open(os.path.join(d, "cari.mtx"), "wb") as g:
        shutil.copyfileobj(f, g)
    return str(d)


@pytest.fixture(scope="session")
def engine():
    import spada_sim_amd as S
    e = S.Engine()
    yield e
    e.close()


def to_oracle(m):
    from oracle import oracle
    return oracle.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)


def assert_parity(c, ref, a=None, b=None, rtol=1e-9):
    """Bit-exact structure; values within rtol relative.  Entries that miss the relative bound must
    meet rtol * sum|a_ik * b_kj| (cancellation); returns how many needed that fallback."""
    from oracle import oracle
    assert tuple(c.shape) == (ref.rows, ref.cols)
    assert np.array_equal(c.indptr, ref.indptr), "C indptr differs"
    assert np.array_equal(c.indices, ref.indices), "C column indices differ"
    err = np.abs(c.data - ref.data)
    bad = err > rtol * np.abs(ref.data)
    nbad = int(bad.sum())
    if nbad:
        assert a is not None, f"{nbad} values outside rtol={rtol}"
        aa = oracle.Csr(a.rows, a.cols, a.indptr, a.indices, np.abs(a.data))
        bb = aa if b is a else oracle.Csr(b.rows, b.cols, b.indptr, b.indices, np.abs(b.data))
        scale = oracle.spgemm_spa(aa, bb).data
        assert np.all(err[bad] <= rtol * scale[bad]), "values outside the cancellation bound"
    return nbad
