import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from os1_amd import api
for d in (2, 3, 4, 5, 6, 8):
    s = api.Stream(500, 1.2, 8, 20, 7, 0, 2, d)
    print('queues', os.environ.get('GPU_MAX_HW_QUEUES'), 'depth', d, 'inflight', s.batches_in_flight())
    s.close()
