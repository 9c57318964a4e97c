from .dataset import Dataset, InteractionIndex  # noqa: F401
