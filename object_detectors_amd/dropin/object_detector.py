"""Registry / factory of detector wrappers (drop-in for the reference's
object_detectors/object_detector.py:4-46).

`@ObjectDetector.register(metadata=dict(frameworks=[..], model_names=[..], tasks=[..],
versions=[..], info={..}))` files a wrapper class under every
(framework, model_name, task, version, json(info)) combination; a duplicate key is a ValueError
(:25-26).  `ObjectDetector(metadata=dict(framework=.., model_name=.., task=.., version=..[, info]),
*args, **kwargs)` looks the key up and returns `wrapper_cls(*args, **kwargs)` -- an instance of the
wrapper, not of ObjectDetector (:31-46); a missing field or unknown key is a ValueError (:40-44).
"""
import itertools
import json

from od_base import ODBase

_FIELDS = ("framework", "model_name", "task", "version")


class ObjectDetector(ODBase):
    _registry = {}

    @classmethod
    def register(cls, metadata):
        plural = {f: metadata.get(f + "s") for f in _FIELDS}
        info = json.dumps(metadata.get("info", {}))

        def decorator(wrapper_cls):
            assert all(plural.values()), "frameworks, model_names, tasks, and versions must be specified."
            for combo in itertools.product(*(plural[f] for f in _FIELDS)):
                key = combo + (info,)
                if key in cls._registry:
                    framework, model_name, task, version = combo
                    raise ValueError(f"Wrapper already registered for version={version}, model_name='{model_name}', "
                                     f"task='{task}', framework='{framework}', info='{info}'.")
                cls._registry[key] = wrapper_cls
            return wrapper_cls

        return decorator

    def __new__(cls, metadata, *args, **kwargs):
        single = tuple(metadata.get(f) for f in _FIELDS)
        if not all(single):
            raise ValueError("Framework, model_name, task, and version must be specified.")
        info = json.dumps(metadata.get("info", {}))
        wrapper_cls = cls._registry.get(single + (info,))
        if wrapper_cls is None:
            framework, model_name, task, version = single
            raise ValueError(f"No class found for version={version}, model_name='{model_name}', task='{task}', "
                             f"framework='{framework}', metadata='{info}'.")
        return wrapper_cls(*args, **kwargs)
