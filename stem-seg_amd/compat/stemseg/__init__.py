"""Reference-compatible namespace for the MI355X hot path (re-exports of stemseg_amd; see ../README.md)."""
