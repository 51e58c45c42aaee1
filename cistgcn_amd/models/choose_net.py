"""Model registry lookup with the contract of the reference's models/choose_net.py:4-11: returns
the network on the current HIP device, raises ValueError for an unknown name."""
import importlib


def choose_net(architecture, opt):
    net = importlib.import_module(__package__)
    registry = {"CISTGCN_0": net.CISTGCN_0, "CISTGCN_eval": net.CISTGCN_eval}
    if architecture not in registry:
        raise ValueError("Network Architecture you are trying to call does not exist in our Repository ;)")
    return registry[architecture](opt.architecture_config, opt.learning_config).cuda()
