"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/cwr_transport.h declares; the product path has no CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'cwr_transport.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cwr_[a-z_0-9]+)\s*\(', text)))


def test_header_and_wrapper_agree():
    import clearwater_riverine_amd as cw
    assert declared_symbols() == sorted(cw.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    import clearwater_riverine_amd as cw
    assert os.path.exists(cw.LIB_PATH), 'build the extension first: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(cw.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} missing from {cw.LIB_PATH}'
    assert cw.load_library().cwr_abi_version() == 7


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, 'include', 'cwr_transport.h')).read()
    assert 'torch' not in text.lower() and 'at::' not in text and 'std::' not in text
    assert 'extern "C"' in text


def test_argument_validation_needs_no_gpu():
    """cwr_create rejects malformed topology before touching the device (ValueError in the wrapper)."""
    import clearwater_riverine_amd as cw
    f1 = np.array([0, 1, 5], dtype=np.int32)          # face 2: face1 = 5 is not a real cell of a 2-cell mesh
    f2 = np.array([1, 2, 3], dtype=np.int32)
    with pytest.raises(ValueError, match='face1'):
        cw.TransportEngine(f1, f2, n_cells=6, n_constituents=1, n_owned=2)
    with pytest.raises(ValueError):
        cw.TransportEngine(f1[:2], f2[:2], n_cells=3, n_constituents=0, n_owned=2)


def test_product_path_fails_loudly_without_a_gpu():
    """No silent CPU fallback: without a GPU, creating an engine raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import clearwater_riverine_amd as cw
    f1 = np.array([0, 0, 1], dtype=np.int32)
    f2 = np.array([1, 2, 3], dtype=np.int32)
    with pytest.raises(RuntimeError):
        cw.TransportEngine(f1, f2, n_cells=4, n_constituents=1)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'clearwater-riverine_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                text = open(os.path.join(dirpath, fn)).read()
                assert 'cwr_oracle' not in text and 'import oracle' not in text, f'{fn} references the oracle'


def test_step_info_layout_matches_the_header(tmp_path):
    """The ctypes StepInfo of engine.py is cwr_step_info of the header, field for field (compiled with gcc as plain C)."""
    import subprocess
    from clearwater_riverine_amd.engine import StepInfo
    src = tmp_path / 'layout.c'
    fields = [name for name, _ in StepInfo._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cwr_transport.h"\nint main(void) {\n'
                   '  printf("%zu", sizeof(cwr_step_info));\n'
                   + ''.join(f'  printf(" %zu", offsetof(cwr_step_info, {f}));\n' for f in fields) + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(StepInfo)
    assert out[1:] == [getattr(StepInfo, f).offset for f in fields]
    # the raw stub shown in INTEGRATION.md declares the same fields
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    for f in fields:
        assert f'"{f}"' in text, f'INTEGRATION.md stub lacks {f}'
