"""Alias package: `src.evaluation.*` re-exports avsum_amd.evaluation.* so that the reference's scripts, which import
`evaluation.*` and `src.evaluation.*`, run unmodified against the MI355X implementation."""
