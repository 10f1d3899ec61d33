"""Build-container only: the weight inventory equals the LIVE reference modules' state_dict()."""
import os

import pytest

import ref_shim


@pytest.mark.reference
@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not mounted (GPU box)")
def test_inventories_equal_reference_state_dicts():
    import yaml
    from megatts2_amd import config as C
    from megatts2_amd import weights
    ref = ref_shim.load()
    from utils.utils import instantiate_class
    root = ref_shim.REFERENCE_ROOT
    G = ref.MegaG.from_hparams(f"{root}/configs/config_gan.yaml")
    plm = instantiate_class((), yaml.safe_load(open(f"{root}/configs/config_plm.yaml"))["model"]["plm"])
    adm = instantiate_class((), yaml.safe_load(open(f"{root}/configs/config_adm.yaml"))["model"]["adm"])
    for mod, inv in ((G, weights.inventory_g(C.production_g())), (plm, weights.inventory_plm(C.production_plm())),
                     (adm, weights.inventory_adm(C.production_adm()))):
        sd = mod.state_dict()
        assert list(sd.keys()) == list(inv.keys())
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(inv[k]), k
