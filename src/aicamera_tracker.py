from . import _pkg

_m = _pkg("cli")
parse_arguments, main = _m.parse_arguments, _m.main

if __name__ == "__main__":
    raise SystemExit(main())
