from ..hyp_model import HypModel
from .base import run_chain


class TransformList(HypModel):
    """Sequence of transforms applied in order (hyperion/transforms/transform_list.py:25-44).
    The whole list is folded into one device plan: consecutive affine maps are multiplied
    together on the host, so the device runs one GEMM per length-normalised segment."""

    _cache_defaults = {"_plan": lambda: None, "_plan_key": lambda: None}

    def __init__(self, transforms, **kwargs):
        super().__init__(**kwargs)
        if not isinstance(transforms, list):
            transforms = [transforms]
        self.transforms = transforms
        self._plan = None
        self._plan_key = None
        if transforms is not None:
            self.update_names()

    def append(self, t):
        self.transforms.append(t)
        if self.name is not None:
            t.name = self.name + "/" + t.name

    def update_names(self):
        if self.name is not None:
            for t in self.transforms:
                t.name = self.name + "/" + t.name

    def predict(self, x):
        if not self.transforms:
            return x
        return run_chain(self.transforms, x, cache_owner=self)

    def save_params(self, f):
        """hyperion/transforms/transform_list.py:58-60."""
        for t in self.transforms:
            t.save_params(f)

    @classmethod
    def load_params(cls, f, config):
        """Every transform of the list from its own group of the file
        (hyperion/transforms/transform_list.py:63-73)."""
        from . import CentWhiten, LDA, LNorm, PCA

        classes = {"CentWhiten": CentWhiten, "LNorm": LNorm, "LDA": LDA, "PCA": PCA}
        transforms = []
        for i in range(len(config["transforms"])):
            config_t = config["transforms"][str(i)]
            if config_t["class_name"] not in classes:
                raise NotImplementedError("transform %s is not part of this package" % config_t["class_name"])
            transforms.append(classes[config_t["class_name"]].load_params(f, config_t))
        return cls(transforms, name=config["name"])

    def get_config(self):
        config = super().get_config()
        config["transforms"] = {i: t.get_config() for i, t in enumerate(self.transforms)}
        return config
