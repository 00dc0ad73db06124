"""Registries mirroring the reference's plugin surface (fvcore Registry semantics):
``REGISTRY.get(name)(cfg, ...)`` builds the component; names are the reference's."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map


META_ARCH_REGISTRY = Registry("META_ARCH")                    # detectron2/modeling/meta_arch/build.py:7
BACKBONE_REGISTRY = Registry("BACKBONE")                      # detectron2/modeling/backbone/build.py:7
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")  # modeling/proposal_generator/build.py
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")                      # modeling/proposal_generator/rpn.py:21
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")      # modeling/anchor_generator.py:13
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")                    # modeling/roi_heads/roi_heads.py:25
