"""tools/ must not rot: every script under tools/ and tools/debug/ parses, every repo module / name it imports exists, and
every `ops.<wrapper>` / `lib.<name>` / `L.<name>` it mentions is an attribute of the current tts_king_amd.ops / lib — checked statically (the scripts
themselves need a GPU), so that a wrapper removed from the ABI fails this CPU suite instead of a profiling session."""
import ast
import glob
import importlib
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "debug", "*.py")))
NEEDS_GPU_AT_IMPORT = {"tools.debug.gemm_tune"}        # runs kernels while being imported


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_tool_script_refers_only_to_what_exists(path):
    sys.path.insert(0, ROOT)
    src = open(path).read()
    tree = ast.parse(src, path)
    missing = []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in ("tts_king_amd", "tools", "oracle", "tests"):
            if node.module in NEEDS_GPU_AT_IMPORT:
                assert os.path.exists(os.path.join(ROOT, *node.module.split(".")) + ".py"), node.module
                continue
            mod = importlib.import_module(node.module)
            for alias in node.names:
                if not hasattr(mod, alias.name):
                    try:
                        importlib.import_module(node.module + "." + alias.name)
                    except ImportError:
                        missing.append("%s.%s" % (node.module, alias.name))
    from tts_king_amd import lib, ops
    for name in set(re.findall(r"\bops\.([A-Za-z_][A-Za-z0-9_]*)", src)):
        if not hasattr(ops, name):
            missing.append("ops." + name)
    declared = set(lib.declared_prototypes())
    for name in set(re.findall(r"\b(ttsk_[a-z0-9_]+)\b", src)):
        family = any(d.startswith(name + "_") for d in declared)           # (a docstring naming a family of entry points: ttsk_gemm_group -> _build, _launch ...)
        if name not in declared and not family and not name.endswith("_set_stamps"):      # (*_set_stamps: hooks of the diagnostic build only)
            missing.append(name)
    assert not missing, "%s refers to %s, which the tree no longer has" % (os.path.relpath(path, ROOT), sorted(missing))
