from .augment import LetterBox  # noqa: F401
