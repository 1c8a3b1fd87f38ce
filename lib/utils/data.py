"""Caller glue kept for import compatibility (reference lib/utils/data.py:3-16)."""
import torch


def data_to_model_device(data, model):
    try:
        device = next(model.parameters()).device
    except StopIteration:
        device = "cpu"
    for k, v in data.items():
        if torch.is_tensor(v):
            data[k] = v.to(device)
    return data
