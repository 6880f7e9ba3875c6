from marl_amd.env.single_state_matrix_game import TwoAgentsMatrixGame, BatchedMatrixGame  # noqa: F401
