from .ssd.make_dataset import SSDDataLoader

__all__ = ["SSDDataLoader"]
