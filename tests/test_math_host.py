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


def test_general_root_finder_of_the_moving_obstacle_environment_matches_the_shim(tmp_path):
    """solve_any6 / poly_max_abs (mplx_poly_dev.h, host build of the device code) == solve() / Primitive1D::max_abs of
    include/mpl_shim -- the statements the compiled-reference checker links -- on 400 000 polynomials of degree 1..5
    (random, lattice coefficients with roots at interval ends, triple roots, nearly quadratic): every root bit-exact."""
    exe = str(tmp_path / "test_poly_solve_host")
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "include", "mpl_shim"), "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tests", "cpp", "test_poly_solve_host.cpp")]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "OK (0 failures)" in out.stdout
