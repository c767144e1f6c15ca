import sys, pytest
sys.path.insert(0, '.')
from sound_event_detection_transformer_amd import ops
for kv in sys.argv[1].split(','):
    if kv:
        k, v = kv.split('=')
        setattr(ops, k, bool(int(v)))
        print('set', k, getattr(ops, k))
sys.exit(pytest.main(['tests/test_model_gpu.py', '-m', 'gpu', '-q', '--x3', '-k', 'g2_g3 and urban', '-x', '--no-header', '-p', 'no:cacheprovider']))
