"""Drop-in for crates/alpharat-mcts-python/python/alpharat_mcts/__init__.py (MI355X backend)."""
from alpharat_amd.mcts import SearchResult, rust_mcts_search

__all__ = ["rust_mcts_search", "SearchResult"]
