"""Opt-in kernel variants behind environment switches (the library reads them once per process): each runs the
tests of its default form in a child process, held to the same bars -- so that no selectable code path is left
without a parity check ("a kernel variant ... that no test reaches")."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    # the bf16-row GEMM with 64-k stages (default 32)
    ({'DODT_FC_BF16_DMA_BK': '64'}, ['tests/test_gpu_heads.py', '-k', 'bf16_rows']),
    # ... with a ring of six stage images (default three)
    ({'DODT_FC_BF16_DMA_RING': '6'}, ['tests/test_gpu_heads.py', '-k', 'bf16_rows']),
    # bf16 heads on float32 activations, rounded on every load (round 3's form), through the whole pair
    ({'DODT_FC_BF16_ROWS': '0'}, ['tests/test_gpu_heads.py', '-k', 'stagewise and bf16']),
    # correlation: 512-lane workgroups with the channels split over half-waves; round 3's two-pass kernel
    ({'DODT_CORR_HALVES': '2'}, ['tests/test_gpu_heads.py', '-k', 'correlation']),
    ({'DODT_CORR_TWO_PASS': '1'}, ['tests/test_gpu_heads.py', '-k', 'correlation']),
    # one work queue per launch instead of one per group of blocks that share an XCD
    # bf16 convs: 16-row tiles also where the map gives fewer than 1.6 items per CU (the small test maps otherwise
    # all take the 8-row tiles)
    ({'DODT_CONV_BF16_MT2': '0'}, ['tests/test_gpu_conv_bf16.py']),
    ({'DODT_CONV_BF16_MT2_RULE': 'rounds'}, ['tests/test_gpu_conv_bf16.py', '-k', 'all_layers']),
    ({'DODT_CONV_BF16_XCD': '0', 'DODT_CONV_F32_XCD': '0'},
     ['tests/test_gpu_conv_bf16.py', 'tests/test_gpu_conv.py', '-k', 'not other_fp32']),
    # the streaming bf16 kernels with shallower rings (the default takes the deepest that fits twice per CU)
    ({'DODT_CONV_BF16_STREAM_LDS': '72'}, ['tests/test_gpu_conv_bf16.py', '-k', 'all_layers']),
    # the tail's elementwise ops as separate launches; the correlation map inside frame 1's tail (round 4's first form)
    ({'DODT_PIPE_FUSED_TAIL': '0', 'DODT_PIPE_CORR_MAP': 'f1'},
     ['tests/test_gpu_heads.py', 'tests/test_gpu_pipeline.py', '-k', 'stagewise or lookahead or pipelined']),
]


@pytest.mark.parametrize('env,args', VARIANTS, ids=[' '.join('%s=%s' % kv for kv in v[0].items()) for v in VARIANTS])
def test_variant_passes_the_default_forms_tests(env, args):
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'] + args,
                       cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert ' passed' in tail and 'no tests ran' not in tail, tail
