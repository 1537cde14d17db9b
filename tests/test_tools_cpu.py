"""The profile post-processing tools (tools/*.py) are part of the measurement chain behind profiles/: keep them runnable. CPU only, tiny
synthetic inputs in the formats rocprofv3 / the test session write."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forward_timeline_collapses_runs_and_starts_at_the_last_patchify(tmp_path):
    d = tmp_path / "trace" / "host"
    d.mkdir(parents=True)
    head = '"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"\n'
    rows = []
    t = 1000

    def k(name, dur, wg, grid):
        nonlocal t
        rows.append(f'"KERNEL_DISPATCH",1,1,1,1,{len(rows)},1,"{name}",0,{t},{t + dur},{wg},1,1,{grid},1,1\n')
        t += dur + 10

    for _ in range(2):  # two forwards: only the second one is reported
        k("(anonymous namespace)::patchify_kernel(void const*, int)", 500, 256, 256 * 16)
        k("void (anonymous namespace)::gemm8_kernel<0, 0, 6>(GemmParams)", 2000, 512, 512 * 652)
        k("void (anonymous namespace)::gemm8_kernel<0, 0, 6>(GemmParams)", 3000, 512, 512 * 652)
        k("void (anonymous namespace)::layernorm_kernel<4>(float const*)", 400, 256, 256 * 100)
    (d / "1_kernel_trace.csv").write_text(head + "".join(rows))
    out = tmp_path / "tl.md"
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "forward_timeline.py"), str(tmp_path / "trace"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    assert "One forward, 4 launches" in text
    assert "| 1 | `gemm8_kernel<0, 0, 6>` | 2 | 652 | 512 | 2.5 | 5.0 |" in text  # two launches of one kernel at one grid: one row
    assert text.count("patchify_kernel") == 1


def test_summarize_parity_groups_by_file_worst_first(tmp_path):
    rec = {"tests/test_a.py::test_x[dtype1-0.02]": 1.3e-2, "tests/test_a.py::test_x[dtype0-0.0001]": 2.5e-5, "tests/test_b.py::test_y": 4e-3}
    src, out = tmp_path / "r.json", tmp_path / "r.md"
    src.write_text(json.dumps(rec))
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "summarize_parity.py"), str(src), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = out.read_text()
    assert text.index("test_x[dtype1-0.02]") < text.index("test_x[dtype0-0.0001]") < text.index("### tests/test_b.py")
    assert "| `test_y` | 0.004 |" in text


def test_every_probe_and_tool_script_compiles_and_resolves_its_repo_imports():
    """two probes still imported a module from its old location. Byte-compile every script under tools/ and check that
    every `from tests...` / `from tools...` / `from muggled_dpt_amd...` import names a module that exists in the repo."""
    import ast, glob, importlib.util, os, py_compile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(repo, "tools", "*.py")) + glob.glob(os.path.join(repo, "tools", "probes", "*.py")))
    assert scripts
    for path in scripts:
        py_compile.compile(path, doraise=True)
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in ("tests", "tools", "muggled_dpt_amd", "oracle"):
                rel = os.path.join(repo, *node.module.split("."))
                assert os.path.exists(rel + ".py") or os.path.isdir(rel), f"{os.path.relpath(path, repo)} imports missing module {node.module}"
