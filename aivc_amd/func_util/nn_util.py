"""Dict helpers with the reference's names (src/func_util/nn_util.py)."""
import torch

from .console_display import print_log_msg


def get_value(key, dic, default_dic):
    """dic[key] if present (and not None) else default_dic[key] (src/func_util/nn_util.py:142-158)."""
    v = dic.get(key)
    if v is None:
        if key in default_dic:
            v = default_dic.get(key)
        else:
            print_log_msg('ERROR', 'get_param', 'key not in default_dic', key)
    return v


def push_dic_to_device(x, device):
    for k in x:
        v = x.get(k)
        if isinstance(v, dict):
            push_dic_to_device(v, device)
        elif isinstance(v, torch.Tensor):
            x[k] = v.to(device, non_blocking=True)
    return x


def push_gop_to_device(x, device):
    for f in x:
        x[f] = push_dic_to_device(x.get(f), device)
    return x


def crop_dic(dic_to_crop, dic_target_size):
    for k in ('y', 'u', 'v'):
        t = dic_target_size.get(k)
        dic_to_crop[k] = dic_to_crop.get(k)[:, :, :t.size()[2], :t.size()[3]]
    return dic_to_crop


def add_dummy_batch_dim(x):
    return x.view(1, *x.shape)


def add_dummy_batch_dim_dic(x):
    for k in ('y', 'u', 'v'):
        x[k] = add_dummy_batch_dim(x.get(k))
    return x


def dic_zeros_like(x):
    return {k: torch.zeros_like(v) for k, v in x.items()}


def convert_tensor_to_dic(x):
    return {'y': x[:, 0:1], 'u': x[:, 1:2], 'v': x[:, 2:3]}
