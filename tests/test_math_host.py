"""CPU: compiles tests/cpp/test_math_host.cpp (the product's f64 math, host build) against the
oracle and runs it: specialised == generic evaluators and product == oracle, bit for bit."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_math_matches_oracle_bit_for_bit(tmp_path):
    from oracle import orc
    orc.build()
    exe = str(tmp_path / "test_math_host")
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__",
           "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_math_host.cpp"),
           os.path.join(ROOT, "oracle", "liborc.so"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "OK (0 failures)" in out.stdout
