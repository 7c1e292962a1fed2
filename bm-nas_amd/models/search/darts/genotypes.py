"""Genotype containers and primitive registries of the BM-NAS search space.

Same module path and names as the reference (models/search/darts/genotypes.py:3-21) —
pickled genotypes (best_genotype.pkl) embed ``models.search.darts.genotypes``, so files
written by either implementation load in the other.
"""
from collections import namedtuple

Genotype = namedtuple('Genotype', 'edges steps concat')
StepGenotype = namedtuple('StepGenotype', 'inner_edges inner_steps inner_concat')

# cell-level edge primitives (reference genotypes.py:6-9)
PRIMITIVES = [
    'none',
    'skip',
]

# edge primitives inside a step node (reference genotypes.py:11-14)
STEP_EDGE_PRIMITIVES = [
    'none',
    'skip',
]

# fusion primitives of a step node (reference genotypes.py:16-21)
STEP_STEP_PRIMITIVES = [
    'Sum',
    'ScaleDotAttn',
    'LinearGLU',
    'ConcatFC',
]
