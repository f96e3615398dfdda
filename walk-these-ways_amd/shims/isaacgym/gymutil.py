def parse_device_str(device_str):
    parts = str(device_str).split(":")
    return parts[0], int(parts[1]) if len(parts) > 1 else 0


def parse_sim_config(sim_cfg, sim_params):
    for k, v in sim_cfg.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                setattr(sim_params.physx, kk, vv)
        else:
            setattr(sim_params, k, v)
