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


def test_cpu_list_parser_behind_the_numa_binding(lib):
    """pcop_bind_thread_near_device reads /sys/bus/pci/devices/<bus id>/local_cpulist: the parser on the forms the kernel writes."""
    from pointcloud_obstacle_processing_amd import capi
    capi.load_library()
    buf = (C.c_int32 * 512)()

    def parse(txt):
        n = lib.pcop_debug_parse_cpulist(txt, buf, 512)
        return n if n < 0 else list(buf[:n])

    assert parse(b"0-3\n") == [0, 1, 2, 3]
    assert parse(b"0-47,96-143\n") == list(range(48)) + list(range(96, 144))
    assert parse(b"5") == [5]
    assert parse(b"1,3,5-6") == [1, 3, 5, 6]
    assert parse(b"") == []
    assert parse(b"3-1") == -1 and parse(b"a-b") == -1 and parse(b"1;2") == -1


def test_numa_binding_without_a_device_changes_nothing(lib):
    if lib.pcop_device_count() > 0:
        pytest.skip("a GPU is present")
    before = os.sched_getaffinity(0)
    lib.pcop_bind_thread_near_device.restype = C.c_int
    assert lib.pcop_bind_thread_near_device(0) == -3
    assert os.sched_getaffinity(0) == before


@pytest.mark.parametrize("flags", [["-DPCOP_REDUCE_AOS=1", "-DPCOP_EMIT_VEC=1", "-DPCOP_DIAG_REUSE_PLAN=1"], ["-DPCOP_DIAG_NO_INDEX_COPY=1"]],
                         ids=["aos+emitvec+reuse", "noidx"])
def test_the_tuning_and_diagnostic_builds_still_parse(flags):
    """The build switches of README.md (tuning / diagnostic builds for tools/ab_libs.sh) are dead text in the default build: a change
    to the kernels around them would otherwise rot them unnoticed.  hipcc -fsyntax-only over both passes, two seconds."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pointcloud_obstacle_processing_amd", "csrc", "pcop_capi.hip")
    out = subprocess.run([hipcc, "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", "-Wno-unused-value", "-Wno-unused-result"] + flags + [src],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and " error" not in out.stderr, out.stderr[-2000:]
