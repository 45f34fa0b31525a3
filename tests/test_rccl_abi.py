"""CPU: the RCCL boundary of the halo exchange, checked against the REAL library before any
multi-GPU hardware sees it.

librfhip.so calls RCCL through dlsym'ed function pointers (reforge_amd/csrc/rf_rccl_abi.h) and the
GPU exchange tests run against a shared-memory double (tests/native/fake_rccl.cpp).  Both could
drift from the real prototypes without anyone noticing on a one-GPU box, so:
  * tests/native/rccl_abi_check.cpp is compiled against the real <rccl/rccl.h>: every function
    type the product binds must be ABI-equivalent to RCCL's own declaration (static_assert), and
    the double is static_asserted against the same types;
  * the real librccl.so.1 must export all eight symbols;
  * the exchange code must close its group on every path."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = "/opt/rocm/lib/llvm/bin/clang++"
INC = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "reforge_amd", "csrc")]


def test_product_types_match_the_real_rccl_header_and_library(tmp_path):
    exe = str(tmp_path / "rccl_abi_check")
    subprocess.check_call([CXX, "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "native", "rccl_abi_check.cpp"), "-o", exe, "-ldl"])
    out = subprocess.check_output([exe, "/opt/rocm/lib/librccl.so.1"]).decode()
    assert len(out.strip().splitlines()) == 8 and "librccl" in out


def test_the_double_implements_the_same_types(tmp_path):
    # syntax + static_asserts only: the double needs libamdhip64 to link, not to type-check
    subprocess.check_call([CXX, "-std=c++17", "-fsyntax-only", "-x", "c++"] + INC + [os.path.join(ROOT, "tests", "native", "fake_rccl.cpp")])


def test_every_symbol_the_product_binds_is_in_the_abi_header():
    src = open(os.path.join(ROOT, "reforge_amd", "csrc", "rf_graph.cpp")).read()
    bound = set(re.findall(r'sym\("(nccl[A-Za-z]+)"\)', src))
    assert bound == {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart",
                     "ncclGroupEnd", "ncclGetErrorString"}
    check = open(os.path.join(ROOT, "tests", "native", "rccl_abi_check.cpp")).read()
    for s in bound:
        assert "decltype(&%s)" % s in check, s


def test_exchange_rows_closes_its_group_on_every_path():
    src = open(os.path.join(ROOT, "reforge_amd", "csrc", "rf_graph.cpp")).read()
    body = src[src.index("rf_status exchange_rows("):src.index("bool exchange_mode(")]
    body = re.sub(r"//[^\n]*", "", body)
    between = body[body.index("GroupStart());") + len("GroupStart());"):body.index("GroupEnd()")]
    assert "return" not in between, "a return between ncclGroupStart and ncclGroupEnd leaves the group open"
    assert "NCCL_TRY" not in between, "NCCL_TRY returns early inside the open group"


def _unique_id_in_a_child(env_extra):
    """rf_comm_unique_id in a fresh process (the library binds RCCL once per process)"""
    code = ("import sys; sys.path.insert(0, %r)\nimport reforge_amd as rf\n"
            "try:\n    uid = rf.Context.unique_id(); print('OK', len(uid), rf.lib().rf_comm_library().decode())\n"
            "except rf.RfError as e:\n    print('ERR', e)\n" % ROOT)
    import sys
    return subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=120).stdout


def test_rf_rccl_library_binds_the_named_build(tmp_path):
    """RF_RCCL_LIBRARY names the RCCL build to bind (a host that has mapped its own librccl.so.1 -- PyTorch -- would otherwise win
    the dlopen by SONAME); a path that cannot be loaded is an error, never a silent fall-back to some other copy."""
    out = _unique_id_in_a_child({"RF_RCCL_LIBRARY": str(tmp_path / "no_such_librccl.so")})
    assert out.startswith("ERR") and "RF_RCCL_LIBRARY" in out, out
    so = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-shared", "-fPIC", "-x", "c++", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp")]
                          + INC + ["-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", so], stderr=subprocess.DEVNULL)
    out = _unique_id_in_a_child({"RF_RCCL_LIBRARY": so})
    assert out.startswith("OK 128") and "libfake_rccl.so" in out, out
