"""CPU-only: .ini / model_spec.json handling of the InferenceEngine facade (no GPU: Init must fail
loudly at the device check, never fall back)."""
import os

import pytest

import inferflow_amd as ia
from inferflow_amd.engine import InferenceEngine, EngineError
from tests import engine_fixtures as fx


def _no_gpu():
    return ia.lib().ifa_device_count() == 0


def test_missing_file_and_section(tmp_path):
    with pytest.raises(EngineError, match="configuration file"):
        InferenceEngine.from_ini(str(tmp_path / "nope.ini"))
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="synthetic")
    with pytest.raises(EngineError, match="Section"):
        InferenceEngine.from_ini(ini, "no_such_section")


@pytest.mark.parametrize("old,new,msg", [
    ("device_weight_data_type = Q4", "device_weight_data_type = Q9", "device_weight_data_type"),
    ("device_kv_cache_data_type = Q8", "device_kv_cache_data_type = int3", "device_kv_cache_data_type"),
    ("models = tiny_test", "models = other", "directory of model"),
    ("model_specification_file = model_spec.json", "model_specification_file = missing.json", "specification file"),
    ("devices = 0", "devices = 0&1;2", "same size"),
    # per-tensor override of the weight type (inference_engine.cc:1664-1690): an unknown type name is refused with the key's name
    ("device_weight_data_type = Q4", "device_weight_data_type = Q4\ndevice_weight_data_type.ffn_w2 = Q7x", "device_weight_data_type.ffn_w2"),
])
def test_ini_errors(tmp_path, old, new, msg):
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="synthetic")
    text = open(ini).read()
    assert old in text
    open(ini, "w").write(text.replace(old, new))
    with pytest.raises(EngineError, match=msg):
        InferenceEngine.from_ini(ini)


def test_bad_spec_json(tmp_path):
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="synthetic")
    open(os.path.join(str(tmp_path), "model_spec.json"), "w").write('{"network_structure": {"normalization_function": "weird"}}')
    with pytest.raises(EngineError, match="normalization_function"):
        InferenceEngine.from_ini(ini)
    open(os.path.join(str(tmp_path), "model_spec.json"), "w").write('{"network_structure": ')
    with pytest.raises(EngineError, match="JSON"):
        InferenceEngine.from_ini(ini)


def test_device_groups_are_validated_before_anything_is_loaded(tmp_path):
    """devices = 0&1;2 -> groups of different sizes (inference_engine.cc:1738-1783); a device named twice; a device
    that does not exist: all refused with the reference's kind of message, on a box with or without GPUs"""
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="synthetic")
    text = open(ini).read()
    open(ini, "w").write(text.replace("devices = 0", "devices = 0&1;2"))
    with pytest.raises(EngineError, match="same size"):
        InferenceEngine.from_ini(ini)
    open(ini, "w").write(text.replace("devices = 0", "devices = 0&977"))
    with pytest.raises(EngineError, match="not available"):
        InferenceEngine.from_ini(ini)


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without a GPU")
def test_init_fails_loudly_without_a_gpu(tmp_path):
    ini, _ = fx.write_model_dir(str(tmp_path), fmt="llama2.c")
    with pytest.raises(EngineError, match="not available"):
        InferenceEngine.from_ini(ini)
