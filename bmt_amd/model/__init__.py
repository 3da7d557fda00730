"""MI355X-native drop-in for the reference's ``model`` package (same module / class names,
constructor and forward signatures, state_dict keys -- SURVEY.md 8b, Appendix A)."""
