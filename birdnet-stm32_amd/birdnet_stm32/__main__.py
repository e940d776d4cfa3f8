"""``python -m birdnet_stm32 <command>`` dispatcher (reference: birdnet_stm32/__main__.py:12-47).

``evaluate`` and ``convert`` (own post-training quantisation, no TensorFlow) exist in this build; the reference's train /
deploy / board-test commands are outside the accelerated path and answer with a pointer to the reference package.
"""

import sys

USAGE = "Usage: birdnet-stm32 {train,convert,evaluate,deploy,board-test}"


def main():
    if len(sys.argv) < 2:
        print(USAGE)
        sys.exit(1)
    command = sys.argv[1]
    sys.argv = [f"birdnet-stm32 {command}"] + sys.argv[2:]
    if command == "evaluate":
        from birdnet_stm32.cli.evaluate import main as run

        run()
    elif command == "convert":
        from birdnet_stm32.cli.convert import main as run

        run()
    elif command in ("train", "deploy", "board-test"):
        print(f"'{command}' is not part of the MI355X hot-path build; use the reference package for it.")
        sys.exit(2)
    else:
        print(f"Unknown command: {command}")
        print(USAGE)
        sys.exit(1)


if __name__ == "__main__":
    main()
