"""Alias package: `src.scripts.*` re-exports avsum_amd.scripts.* so that the reference's scripts, which import
`scripts.*` and `src.scripts.*`, run unmodified against the MI355X implementation."""
