"""Model resolution shared by the CLI scripts: '../models/<name>/0_model.pt' (a reference pickle) when
present, else the synthetic stand-in (weights are not in the snapshot)."""
import os

import torch


def resolve_device(cpu_flag):
    if cpu_flag:
        raise SystemExit('[ERROR] --cpu: aivc_amd has no CPU execution path (HIP kernels only); the CPU '
                         'restatement lives in oracle/ and is test infrastructure')
    if not torch.cuda.is_available():
        raise SystemExit('[ERROR] no GPU visible: aivc_amd needs an MI355X (no CPU fallback)')
    return torch.device('cuda:0')


def get_model(name, device, models_dir='../models'):
    import aivc_amd
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
        return model.to(device).eval()
    print('[INFO] assets absent: %s/0_model.pt not found, using the synthetic random-init model' % path)
    model = synth.make_model(device=device)
    synth.calibrate_operating_point(model, device)
    return model
