"""`model.layer` of the reference tree (model/layer.py) -> the MI355X classes."""
from objectdetection_amd.model.layer import cbs, concat, elan, elan_h, mp, sppcspc, head, decode, deploy  # noqa: F401
