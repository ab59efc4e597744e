from .raymarching import *  # noqa: F401,F403  (same re-export as the reference's raymarching/__init__.py)
