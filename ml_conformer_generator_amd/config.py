"""Framework-level constants of the hot path.

Values must equal the reference's `src/mlconfgen/utils/config.py:3-32`; they are
data (model hyper-parameters of the published checkpoints), not code.
"""

# GCN pad width (atoms) and number of bond classes  (config.py:3-4)
DIMENSION = 42
NUM_BOND_TYPES = 5

# Context (principal moments of inertia) normalisation  (config.py:5-8)
CONTEXT_NORMS = {
    "mean": [105.0766, 473.1938, 537.4675],
    "mad": [52.0409, 219.7475, 232.9718],
}

# class index -> element symbol  (config.py:9-18)
ATOM_DECODER = {0: "C", 1: "N", 2: "O", 3: "F", 4: "P", 5: "S", 6: "Cl", 7: "Br"}
# class index -> atomic number (sorted PERMITTED_ELEMENTS, config.py:20-29,
# molgraph.py:10 `elements_decoder`)
ATOMIC_NUMBERS = (6, 7, 8, 9, 15, 16, 17, 35)
PERMITTED_ELEMENTS = ATOMIC_NUMBERS

# generated molecule size range (heavy atoms)  (config.py:31-32)
MIN_N_NODES = 15
MAX_N_NODES = 39

# ---- architecture hyper-parameters hard-coded in the reference constructor
# (conformer_generator.py:67-88)
EGNN_HIDDEN = 420          # hidden_nf
EGNN_IN_NODE_NF = 12       # 8 atom classes + 1 time + 3 context
EGNN_N_BLOCKS = 9          # e_block_0..8
EGNN_NORM_FACTOR = 100.0   # unsorted_segment_sum normalisation (egnn.py:435)
N_ATOM_CLASSES = 8
N_DIMS = 3
NOISE_PRECISION = 1e-5
NORM_VALUES = (1.0, 9.0)   # equivariant_diffusion.py:149-152

GCN_HIDDEN = 2048
GCN_EMBED = 64
GCN_NUM_EMBEDDINGS = 36
