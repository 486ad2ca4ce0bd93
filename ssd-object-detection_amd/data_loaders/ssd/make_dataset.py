"""SSDDataLoader with the reference's contract (data_loaders/ssd/make_dataset.py:15-74 of the reference):

    SSDDataLoader(dataset_root, dataset="coco", shuffle=True, mini_batch=0)
    .get_dataset() -> (train, val) iterables of (image f32[300,300,3] RGB in [0,1], cls f32[n], box f32[n,4])
                      with boxes (cx, cy, w, h) normalised to [0,1]
    .get_names_and_colors() -> (names, colors)

The model consumes only this contract.  Sources:
  dataset="synthetic": COCO-shaped random samples; no files, no network.
  dataset=<reader>:    any object with the reference COCO reader's contract (data_loaders/coco/make_dataset.py:32-36,
                       100-134 of the reference): get_dataset() -> (train, val) iterables of (decoded image [H,W,3] --
                       uint8, or float in [0,1] as `imread / 255` gives it --, cls [n], box [n,4] = (cx, cy, w, h) in
                       PIXELS of that image), optionally get_names_and_colors().  This is the `_coco2ssd` seam (:37-46): the
                       resize to 300x300, the /255 and the box normalisation run on the device (ssd_image_resize_prep,
                       ssd_box_prep) when SSDObjectDetectionModel.get_train_set batches such samples (which is what
                       train() does); iterated directly, such a split yields the decoded samples (RawSplit), not resized
                       images: there is no host-side resize in this package.
  dataset="coco":      needs pycocotools + scikit-image + the COCO files (none of which exist in this image);
                       the annotation/IO layer itself is outside this build's scope: wrap a
                       COCO reader of your own as dataset=<reader>.
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


class RawSplit:
    """One split of a reader-contract source, marked for device-side preprocessing: samples are (image uint8 [H,W,3],
    cls f32 [n], box f32 [n,4] = COCO [x, y, w, h] top-left pixels) -- what SSDObjectDetectionModel.make_batch_raw takes."""
    raw = True

    def __init__(self, source, limit=0):
        self._source, self._limit = source, int(limit)

    def __iter__(self):
        for i, (image, cls, box) in enumerate(self._source):
            if self._limit and i >= self._limit:
                break
            image = np.asarray(image)
            if image.ndim == 2:                                          # grey image: reference coco/make_dataset.py:129-130
                image = np.stack([image, image, image], axis=2)
            if image.dtype != np.uint8:                                  # imread / 255 (reference :117): exact inverse
                image = np.rint(np.asarray(image, np.float64) * 255.0).astype(np.uint8)
            box = np.asarray(box, np.float32).reshape(-1, 4).copy()
            box[:, :2] -= box[:, 2:] / np.float32(2.0)                   # centre (reference :132) back to top-left
            yield np.ascontiguousarray(image[..., :3]), np.asarray(cls, np.float32), box


class SSDDataLoader:
    def __init__(self, dataset_root, dataset="coco", shuffle=True, mini_batch=0):
        self._train_resize = (300, 300)
        if not isinstance(dataset, str):                                 # a reader object: the _coco2ssd seam
            if not hasattr(dataset, "get_dataset"):
                raise ValueError
            train, val = dataset.get_dataset()
            self._train_set, self._val_set = RawSplit(train, mini_batch), RawSplit(val, 0)
            if hasattr(dataset, "get_names_and_colors"):
                self._names, self._colors = dataset.get_names_and_colors()
            else:
                self._names = ["class_%02d" % i for i in range(COCO_CLASS_COUNT)]
                self._colors = [[128, 128, 128]] * COCO_CLASS_COUNT
            return
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
