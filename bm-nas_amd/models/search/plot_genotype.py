"""Genotype plotting hook (reference models/search/plot_genotype.py:13-21).  Rendering needs the
graphviz `dot` binary and the reference's visualize module; both are out of scope, so plotting is
attempted only when they import and is otherwise skipped (the search itself never depends on it)."""


class Plotter():
    def __init__(self, args):
        self.args = args
        try:
            from .darts.visualize import plot
            self._plot = plot
        except Exception:
            self._plot = None

    def plot(self, genotype, file_name, task=None):
        if self._plot is None:
            return
        try:
            self._plot(genotype, file_name, self.args, task)
        except Exception:
            pass
