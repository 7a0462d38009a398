"""Scalar logging stand-in for the reference's wandb logger (reference logger.py:10-61): one JSON
line per call on stdout (or a file).  Plots / WaveGlow audio of the reference are out of scope."""
import json
import sys

_sink = sys.stdout


def set_sink(fileobj):
    global _sink
    _sink = fileobj


def _scalar(v):
    try:
        return float(v)
    except (TypeError, ValueError):
        return str(v)


def log_values(step, **kwargs):
    _sink.write(json.dumps(dict(step=step, **{k: _scalar(v) for k, v in kwargs.items() if v is not None})) + '\n')


def log_validation(val_mel_loss, val_gate_loss, val_attn_loss, iteration, **_unused):
    log_values(iteration, validation_mel_loss=val_mel_loss, validation_gate_loss=val_gate_loss,
               validation_attention_loss=val_attn_loss)
