"""ASan + UBSan over the product's HOST code: host/lm.cpp, host/closedform.cpp, csrc/kdvisit.h and the oracle they are
checked against, compiled together with tests/sanitize_harness.cpp by g++ -fsanitize=address,undefined -fno-sanitize-recover=undefined and
run as a plain executable (no GPU, no HIP runtime).  Any out-of-bounds access, use-after-free, signed overflow, misaligned or null access
aborts the harness; it also asserts that the host solve agrees with the oracle's for every parameterization / cost / loss."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_harness")
    src = [os.path.join(ROOT, "tests", "sanitize_harness.cpp"), os.path.join(ROOT, "mv-lm-icp_amd", "host", "lm.cpp"),
           os.path.join(ROOT, "mv-lm-icp_amd", "host", "closedform.cpp"), os.path.join(ROOT, "oracle", "oracle.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-Wno-unknown-pragmas", "-o", exe] + src, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    txt = out.stdout.decode()
    assert out.returncode == 0 and "SANITIZE_HARNESS_OK" in txt, txt[-4000:]
    assert "runtime error" not in txt and "AddressSanitizer" not in txt, txt[-4000:]
