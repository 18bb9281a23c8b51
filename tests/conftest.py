import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_sessionstart(session):
    """A fresh checkout has no built library (artefacts are not in the history): build it once, like
    `__graft_entry__.build()` does (hipcc cross-compiles without a GPU; about two minutes)."""
    lib = os.path.join(ROOT, "theia_amd", "lib", "libtheia_hip.so")
    if os.path.exists(lib):
        return
    import shutil
    import subprocess

    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        return  # the tests that need the library will say so
    jobs = str(min(8, os.cpu_count() or 1))
    print(f"\n[conftest] {lib} is missing: building it (make -C theia_amd/csrc -j {jobs})", flush=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "theia_amd", "csrc"), "-j", jobs, "ARCH=gfx950"], check=True,
                   stdout=subprocess.DEVNULL)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running statistical test")


@pytest.fixture
def oracle_backend():
    """Installs the CPU-oracle stand-in as theia_amd's backend for one test."""
    from oracle_backend import OracleBackend
    from theia_amd import _backend

    prev = _backend._backend
    be = OracleBackend(threads=min(8, os.cpu_count() or 1))
    _backend._backend = be
    yield be
    _backend._backend = prev


@pytest.fixture(scope="session")
def hip():
    """The real HIP backend (fails loudly when libtheia_hip.so or the GPU is missing)."""
    from theia_amd import _backend

    return _backend.HipBackend()


@pytest.fixture
def hip_backend(hip):
    from theia_amd import _backend

    prev = _backend._backend
    _backend._backend = hip
    yield hip
    _backend._backend = prev


BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    """runs a test once on the CPU oracle stand-in and once (-m gpu) on the HIP library"""
    from theia_amd import _backend

    prev = _backend._backend
    if request.param == "oracle":
        from oracle_backend import OracleBackend

        be = OracleBackend(threads=min(8, os.cpu_count() or 1))
    else:
        be = _backend.HipBackend()
    _backend._backend = be
    yield request.param
    _backend._backend = prev
    if request.param == "oracle":
        be.close()
