"""Import-path alias of the reference tree: `model.layer` and `model.yolov7` resolve to the MI355X
classes of objectdetection_amd.model.  The reference instantiates the network through
`exec('from model.yolov7 import yolov7')` (train_class.py:105-106) and its checkpoints pickle whole
module objects (train_class.py:262-276), so these module paths and class names are part of the on-disk
format: a reference checkpoint unpickles into the classes of this build."""
