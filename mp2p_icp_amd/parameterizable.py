"""Formula-capable parameters (Parameterizable.h:169-184, Parameterizable.cpp:47-136).

A parameter declared through declare_parameter_{req,opt} may be a number or a string
expression over runtime variables (e.g. "MATCH_THRESHOLD*2.0", "ICP_ITERATION<5 ? 2.0 : 1.0"
is NOT supported -- only arithmetic and the functions of `math`); it is re-evaluated by
ParameterSource.realize() before every ICP iteration (ICP.cpp:131-134).
checkAllParametersAreRealized() guards use (Matcher_Points_DistanceThreshold.cpp:55)."""
import math

_SAFE = {k: getattr(math, k) for k in ("sqrt", "exp", "log", "sin", "cos", "tan", "pow", "pi",
                                       "fabs", "floor", "ceil")}
_SAFE.update({"abs": abs, "min": min, "max": max})


class ParameterSource:
    def __init__(self):
        self.variables = {}
        self._attached = []

    def updateVariable(self, name, value):
        self.variables[name] = float(value)

    def attach(self, obj):
        if obj not in self._attached:
            self._attached.append(obj)
        obj._source = self

    def realize(self):
        for obj in self._attached:
            obj._realize(self.variables)


class Parameterizable:
    def __init__(self):
        self._declared = {}  # name -> (expression or value, cast)
        self._realized = set()
        self._source = None

    def attachToParameterSource(self, src):
        src.attach(self)

    def attachedSource(self):
        return self._source

    def _declare(self, name, value, cast):
        self._declared[name] = (value, cast)
        if not isinstance(value, str):
            setattr(self, name, cast(value))
            self._realized.add(name)
        else:
            try:  # constant expressions are realised immediately
                setattr(self, name, cast(eval(value, {"__builtins__": {}}, dict(_SAFE))))
                self._realized.add(name)
            except Exception:
                self._realized.discard(name)

    def declare_parameter_req(self, params, name, cast=float):
        if params is None or name not in params:
            raise KeyError(f"Required parameter `{name}` not an existing key in dictionary.")
        self._declare(name, params[name], cast)

    def declare_parameter_opt(self, params, name, cast=float):
        if params is not None and name in params:
            self._declare(name, params[name], cast)
        else:
            self._realized.add(name)

    def _realize(self, variables):
        for name, (value, cast) in self._declared.items():
            if isinstance(value, str):
                env = dict(_SAFE)
                env.update(variables)
                setattr(self, name, cast(eval(value, {"__builtins__": {}}, env)))
                self._realized.add(name)

    def checkAllParametersAreRealized(self):
        missing = [n for n in self._declared if n not in self._realized]
        if missing:
            raise RuntimeError(f"Parameters not realized (attach a ParameterSource and call "
                               f"realize()): {missing}")
