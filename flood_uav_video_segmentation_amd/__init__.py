"""MI355X-native key-frame segmentation + flow-interpolation hot path (drop-in for the reference's
flow/model.py::FlowModel and model/{pspnet,deeplabv3,wrapper}.py network wrappers)."""

__all__ = ["ops", "_lib"]
