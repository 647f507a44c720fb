"""tests/c_abi_smoke.c — the C ABI driven from plain C through include/nufft_mi355x.h alone (built by
__graft_entry__.build()).  Without a GPU: struct sizes, a host-only plan, error codes; with one: create -> set_points ->
exec_type1 -> exec_type2 -> destroy, checked in C against the direct sums."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c_abi_smoke")


def _exe():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()
    if not os.path.exists(EXE):
        pytest.skip("tests/c_abi_smoke could not be built on this machine (no C compiler / ROCm headers)")
    return EXE


def test_c_abi_from_plain_c_host_only():
    r = subprocess.run([_exe(), "--host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout


@pytest.mark.gpu
def test_c_abi_from_plain_c_on_the_gpu():
    r = subprocess.run([_exe()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "type-1 rel-L2" in r.stdout
