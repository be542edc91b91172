"""`model.yolov7` of the reference tree (model/yolov7.py) -> the MI355X class."""
from objectdetection_amd.model.yolov7 import yolov7  # noqa: F401
