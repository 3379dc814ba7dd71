"""The package installs under an importable name: a wheel built from this tree (setup.cfg maps the directory
`nano-vllm_amd/` to the package `nanovllm_amd`) imports from a clean directory without the root shim, ships
the HIP library, and that library exports the C ABI."""
import os
import shutil
import subprocess
import sys
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wheel_imports_without_the_root_shim(tmp_path):
    out = tmp_path / "whl"
    r = subprocess.run([sys.executable, "-m", "pip", "wheel", ROOT, "--no-deps", "--no-build-isolation", "-q",
                        "-w", str(out)], capture_output=True, text=True, cwd=str(tmp_path))
    for leftover in ("build/lib", "build/bdist.linux-x86_64", "nanovllm_amd.egg-info"):   # the in-tree build's litter
        shutil.rmtree(os.path.join(ROOT, leftover), ignore_errors=True)
    assert r.returncode == 0, r.stderr[-2000:]
    wheels = [f for f in os.listdir(out) if f.endswith(".whl")]
    assert len(wheels) == 1 and wheels[0].startswith("nanovllm_amd-")
    site = tmp_path / "site"
    with zipfile.ZipFile(out / wheels[0]) as z:
        names = z.namelist()
        z.extractall(site)
    assert "nanovllm_amd/lib/libnanovllm_hip.so" in names, "build the library first: make -C nano-vllm_amd/csrc"
    assert "nanovllm_amd/layers/attention.py" in names and "nanovllm_amd/engine/scheduler.py" in names
    code = ("import nanovllm_amd, os; from nanovllm_amd import _lib; "
            "from nanovllm_amd.layers.attention import Attention; from nanovllm_amd.layers.layernorm import RMSNorm; "
            "from nanovllm_amd.layers.rope import get_rope; "
            "assert os.path.dirname(nanovllm_amd.__file__).startswith(%r), nanovllm_amd.__file__; "
            "assert _lib.lib().nv_abi_version() == _lib.ABI_VERSION; print('ok')" % str(site))
    env = dict(os.environ, PYTHONPATH=str(site))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
