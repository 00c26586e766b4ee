"""Model resolution shared by the CLI scripts: '../models/<name>/0_model.pt' (a reference pickle) when
present, else the synthetic stand-in (weights are not in the snapshot)."""
import os

import torch


def resolve_device(cpu_flag):
    """cuda:0, or cuda:LOCAL_RANK when the script runs as one rank of a `torch.distributed.run` job: intra-period
    units are then sharded over the ranks (aivc_amd/parallel.py; one process per GPU, RCCL), rank 0 reads the
    results, writes the files and prints -- the other ranks stay silent."""
    if cpu_flag:
        raise SystemExit('[ERROR] --cpu: aivc_amd has no CPU execution path (HIP kernels only); the CPU '
                         'restatement lives in oracle/ and is test infrastructure')
    if not torch.cuda.is_available():
        raise SystemExit('[ERROR] no GPU visible: aivc_amd needs an MI355X (no CPU fallback)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return torch.device('cuda:0')
    import torch.distributed as dist
    local = 0 if os.environ.get('AIVC_SINGLE_DEVICE') else int(os.environ.get('LOCAL_RANK', '0'))  # (test aid: all ranks on cuda:0)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if not dist.is_initialized():
        from . import parallel
        parallel.init_process_group(dev)  # RCCL bound to this rank's GPU, finite timeout, watchdog
    if dist.get_rank() != 0:
        from .func_util import console_display
        console_display.FLAG_QUIET = True
    return dev


def get_model(name, device, models_dir=None):
    import aivc_amd
    # the reference runs from src/ and reads ../models/<name>/0_model.pt (src/encode.py:101-103); AIVC_MODELS_DIR moves it
    models_dir = models_dir or os.environ.get('AIVC_MODELS_DIR', '../models')
    from aivc_amd import synth
    from aivc_amd.model_mngt.model_management import load_model
    path = os.path.join(models_dir, name)
    if os.path.isfile(os.path.join(path, '0_model.pt')):
        cwd = os.getcwd()
        os.chdir(path)
        try:
            model = load_model(prefix='0_', on_cpu=True)
        finally:
            os.chdir(cwd)
        from aivc_amd import parallel
        return parallel.broadcast_model(model.to(device).eval())
    from aivc_amd import parallel
    if parallel.rank_world()[0] == 0:
        print('[INFO] assets absent: %s/0_model.pt not found, using the synthetic random-init model' % path)
    model = synth.make_model(device=device)
    synth.calibrate_operating_point(model, device)
    return parallel.broadcast_model(model)  # (no-op outside a distributed job)
