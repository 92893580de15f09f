"""go_ai -- MI355X-native batched MCTS leaf-expansion path behind the reference's
go_ai/search, go_ai/policies, go_ai/models and go_ai/game call surface.

Importing the package does not touch the GPU; the native library is loaded by
`go_ai._hip` on first use and there is no CPU fallback.
"""
__all__ = ["engine", "search", "policies", "models", "game"]
