"""rtk_amd: MI355X-native rtk hot path. `rtk_amd.api` binds librtk_amd.so (ctypes); `synth` holds the
deterministic benchmark scenes; `shard` the multi-GPU ray-range logic. Importing the package loads nothing."""
