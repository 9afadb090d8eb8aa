"""The product's Python path (DeviceModel, State, Contacts, CollisionPipeline, the solver classes, __graft_entry__.smoke) on a
machine without a GPU: a subprocess loads tests/emu/emu_plugin.py, which redirects "cuda" tensors to host memory and points
the product loader at the emulated kernel library, then runs a few of the `-m gpu` test files and smoke() unchanged.
tests/emu/run_gpu_tests_emulated.py does the same for the whole GPU suite (slow; a pre-flight check, not part of this suite)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, "tests")
ENV = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(TESTS, "emu"), ROOT, os.environ.get("PYTHONPATH", "")]))


def test_gpu_test_files_dry_run(oracle_lib):
    files = ["test_xpbd_joint_recovery.py", "test_kinematics.py", "test_gpu_parity_semi_implicit.py", "test_body_force.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "emu_plugin", "-m", "gpu", "-q", "-x", *files], cwd=TESTS, env=ENV,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_smoke_dry_run(oracle_lib):
    r = subprocess.run([sys.executable, "-c", "import emu_plugin\nimport __graft_entry__ as g\ng.smoke()"], cwd=ROOT, env=ENV,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "[smoke] ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_mesh_on_ground_pipeline_dry_run(oracle_lib):
    """Triangle meshes on the ground plane through CollisionPipeline.collide (the vertex leg as pair kind 2 of the SDF leg): the row
    and matching tests of tests/test_gpu_mesh_plane_pipeline.py on the emulated library -- builder, pair routing, the tile's MESH-as-local-AABB
    shape, nt_mesh_plane_pairs, nt_sdf_rows_finalize -- against the checker chain.  (The settle test of that file takes half an hour
    of emulation and stays with the device.)"""
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "emu_plugin", "-m", "gpu", "-q", "-x", "test_gpu_mesh_plane_pipeline.py",
                        "-k", "rows_of_meshes or matching"], cwd=TESTS, env=ENV, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "4 passed" in r.stdout


def test_unreduced_sdf_rows_dry_run(oracle_lib):
    """CollisionPipeline(reduce_contacts=False) with mesh-SDF pairs (the keep-all last stage of the staged narrow phase) on the
    emulated library against the checker's unreduced contacts: tests/test_gpu_sdf_pipeline.py's test, unchanged."""
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "emu_plugin", "-m", "gpu", "-q", "-x", "test_gpu_sdf_pipeline.py",
                        "-k", "unreduced"], cwd=TESTS, env=ENV, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "1 passed" in r.stdout


def test_triangle_leg_pipeline_dry_run(oracle_lib):
    """Convex shapes on a triangle-mesh terrain and on a heightfield through CollisionPipeline.collide (the triangle leg as pair kind 3
    of the SDF leg): the row tests of tests/test_gpu_mesh_triangle_pipeline.py on the emulated library -- builder (nt.Heightfield, hull
    tables, block bounds), pair routing on the host and in nt_model_create, nt_mesh_triangle_pairs, nt_sdf_rows_finalize with the
    partner's effective radius -- against the checker chain.  (The settle tests of that file stay with the device.)"""
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "emu_plugin", "-m", "gpu", "-q", "-x", "test_gpu_mesh_triangle_pipeline.py",
                        "-k", "rows_of_primitives or rows_of_shapes_on_a_heightfield"], cwd=TESTS, env=ENV, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "4 passed" in r.stdout
