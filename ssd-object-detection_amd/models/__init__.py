from .ssd_model import SSDObjectDetectionModel

__all__ = ["SSDObjectDetectionModel"]
