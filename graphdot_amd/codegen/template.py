"""String templates with ``${name}`` / ``${name<sep>}`` / ``?{pyexpr}``
placeholders.

Same placeholder grammar as the reference (``graphdot/codegen/template.py:9-59``)
so that kernel templates and ``repr`` strings render identically:

* ``${name}``      -> ``str(value)``
* ``${name<sep>}`` -> ``<sep>.join(map(str, value))`` when value is a list/tuple
* ``?{pyexpr}``    -> ``true`` / ``false`` (evaluated in :py:meth:`context`)
"""
import os
import re
from contextlib import contextmanager

_cond = re.compile(r'\?{([^}]+)}')


class Template:

    def __init__(self, template, escape=True):
        if os.path.isfile(template):
            with open(template) as f:
                template = f.read()
        self.template = template
        self.escape = escape

    @contextmanager
    def context(self, **scope):
        """Resolve all ``?{expr}`` switches against `scope`; yields a new
        template, the original is left untouched."""
        def decide(m):
            return 'true' if eval(m.group(1), dict(scope)) else 'false'
        yield Template(_cond.sub(decide, self.template), self.escape)

    def render(self, **substitutions):
        text = self.template
        # longest names first so that ${ab} is never clobbered by ${a...}
        for name in sorted(substitutions, key=lambda s: (-len(s), s)):
            value = substitutions[name]
            if isinstance(value, (list, tuple)):
                parts = [str(v) for v in value]
                text = re.sub(r'\${%s([^}]*)}' % re.escape(name),
                              lambda m: m.group(1).join(parts), text)
            else:
                value = str(value)
                if self.escape is False:
                    value = value.replace('\\', r'\\')
                # a function as `repl` keeps backslashes in `value` literal
                # unless the caller asked for re.sub escape processing
                if self.escape is False:
                    text = re.sub(r'\${%s}' % re.escape(name), value, text)
                else:
                    text = re.sub(r'\${%s}' % re.escape(name),
                                  lambda m, v=value: v, text)
        return text
