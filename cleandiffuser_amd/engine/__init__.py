"""gfx950 execution engine: plan compiler, network->program compiler, C-ABI binding, dispatch rules."""
