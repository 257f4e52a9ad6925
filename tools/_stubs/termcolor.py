"""Stand-in for the cosmetic `termcolor` dependency of the reference's display module
(SURVEY.md 8c): used only by tools/gen_golden.py in the build container."""


def colored(text, *args, **kwargs):
    return text
