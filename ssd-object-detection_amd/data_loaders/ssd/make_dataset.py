"""SSDDataLoader with the reference's contract (data_loaders/ssd/make_dataset.py:15-74 of the reference):

    SSDDataLoader(dataset_root, dataset="coco", shuffle=True, mini_batch=0)
    .get_dataset() -> (train, val) iterables of (image f32[300,300,3] RGB in [0,1], cls f32[n], box f32[n,4])
                      with boxes (cx, cy, w, h) normalised to [0,1]
    .get_names_and_colors() -> (names, colors)

The model consumes only this contract.  Two sources are provided:
  dataset="synthetic": COCO-shaped random samples; no files, no network.
  dataset="coco":      needs pycocotools + scikit-image + the COCO files (none of which exist in this image);
                       the annotation/IO layer itself is outside this build's scope.
Any other name raises ValueError, as the reference does (:33)."""
import numpy as np

from ..synthetic import synth_gt, synth_image

COCO_CLASS_COUNT = 80


class _SyntheticSplit:
    def __init__(self, first, count, shuffle, size):
        self.first, self.count, self.shuffle, self.size = first, count, shuffle, size
        self._epoch = 0

    def __len__(self):
        return self.count

    def _sample(self, i):
        cls, box = synth_gt(self.first + i)
        return synth_image(self.first + i, self.size), cls, box

    def lazy(self):
        """One pass in iteration order as zero-argument callables: a consumer that keeps only a shard of the samples
        (data-parallel ranks, models/ssd_model.py:get_train_set) pays for those only.  Same order on every rank."""
        order = np.arange(self.count)
        if self.shuffle:
            np.random.default_rng(977 + self._epoch).shuffle(order)
        self._epoch += 1
        for i in order:
            yield lambda i=int(i): self._sample(i)

    def __iter__(self):
        for thunk in self.lazy():
            yield thunk()


class SSDDataLoader:
    def __init__(self, dataset_root, dataset="coco", shuffle=True, mini_batch=0):
        self._train_resize = (300, 300)
        name = dataset.lower()
        if name == "synthetic":
            n_train = int(mini_batch) if mini_batch else 10000
            self._train_set = _SyntheticSplit(0, n_train, shuffle, self._train_resize[0])
            self._val_set = _SyntheticSplit(1 << 20, max(1, n_train // 10), False, self._train_resize[0])
            self._names = ["class_%02d" % i for i in range(COCO_CLASS_COUNT)]
            rng = np.random.default_rng(0)
            self._colors = [rng.integers(80, 240, 3).tolist() for _ in range(COCO_CLASS_COUNT)]
        elif name == "coco":
            try:
                import pycocotools.coco  # noqa: F401
                import skimage.io  # noqa: F401
            except ImportError as e:
                raise ImportError("dataset='coco' needs pycocotools and scikit-image plus the COCO files under %r; "
                                  "use dataset='synthetic' here" % (dataset_root,)) from e
            raise NotImplementedError("COCO annotation/IO layer is out of scope of this build (SURVEY.md section 2a)")
        else:
            raise ValueError

    def get_dataset(self):
        return self._train_set, self._val_set

    def get_names_and_colors(self):
        return self._names, self._colors
