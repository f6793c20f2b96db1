"""A short run of the random-scene soaks (tools/soak_path.py, tools/soak_sppm.py, tools/soak_film.py) inside the suite: Path / Whitted / SPPM on the GPU against
the oracle on scenes nobody designed — Cornell walls plus random matte / plastic / mirror / glass triangles and spheres, point or spot light.
The long runs are recorded under profiles/r2/r2t_soak_*."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,args", [("soak_path.py", ["--scenes", "8", "--seed", "21"]), ("soak_sppm.py", ["--scenes", "10", "--seed", "22"]),
                                       ("soak_film.py", ["--cases", "12", "--seed", "23"])])
def test_random_scene_soak(T, tool, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 with a mismatch" in r.stdout
