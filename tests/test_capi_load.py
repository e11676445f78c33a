"""CPU-side checks of the drop-in boundary: libpcop.so builds, loads and exports every symbol
include/pcop.h declares; without a GPU it refuses to create a context (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pointcloud_obstacle_processing_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return capi.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcop.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcop_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    from pointcloud_obstacle_processing_amd import capi
    names = declared_symbols()
    assert len(names) >= 20
    for nm in names:
        assert hasattr(lib, nm), "libpcop.so does not export %s" % nm
    assert sorted(capi.EXPORTS) == names


def test_params_struct_is_32_bytes():
    from pointcloud_obstacle_processing_amd import capi
    assert C.sizeof(capi.Params) == 32


def test_status_strings(lib):
    assert lib.pcop_status_string(0) == b"ok"
    assert b"CPU" in lib.pcop_status_string(-3)


def test_no_cpu_fallback(lib):
    """Without a HIP device the product path must fail loudly, never compute on the host."""
    if lib.pcop_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.pcop_create(C.byref(h), 0, 1000) == -3
    assert lib.pcop_create(C.byref(h), -1, 1000) == -3
    assert not h.value


def test_product_does_not_import_oracle():
    """The package and the C ABI sources never reference oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "pointcloud_obstacle_processing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pcop_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_root_tools_do_not_touch_the_oracle():
    """tools/ at the repository root is oracle-free (profile reduction, sweeps); what needs the oracle lives in tests/tools."""
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            text = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            assert "pcop_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f
